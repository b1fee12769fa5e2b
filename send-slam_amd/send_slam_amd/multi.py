"""Multi-GPU shapes of the ORB path; one process per GPU over
torch.distributed ("nccl" is RCCL over xGMI on ROCm; "gloo" for CPU rehearsal).

Nothing in the reference corresponds to this (it has one camera, one backend process, one TCP
link: README.md:5, application.ex:80); the three shapes come from BASELINE.json's configs:

  config 3  independent cameras     shard_cameras(): camera ids per rank, NO collective
  config 4  stereo, 2 GPUs          stereo_exchange(): all_gather of fixed-size descriptor
                                    blocks (kp_capacity x 32 B per rank) + counts, then each
                                    rank matches its own eye against the peer's
  config 5  loop closure, N GPUs    loop_closure_query(): keyframe DB partitioned in contiguous
                                    slabs; query broadcast; every rank reports raw
                                    (d1, j1, d2) per query descriptor for its slab; all_gather
                                    of N x nq x 8 B; element-wise fold in rank order (ties keep
                                    the lower global index; second best = min over the losers'
                                    d1 and everyone's d2); then the threshold / ratio test

All messages are <= 64 KB per rank: latency-bound on xGMI, so one all_gather per exchange and
no bucketing.  The fold is the same rule the match kernel uses across its train chunks
(csrc/ss_kernels.hip merge_partial).

On the GPU the whole query is three launches behind the C ABI and no torch arithmetic:
ss_match_partial_device (raw local match -> 8-byte records with global rows), one all_gather of
N x nq x 8 B, ss_match_fold_device (k_match_merge over the gathered records + acceptance test).
Stream order: every torch / RCCL call of a query runs with the context's own HIP stream as torch's
current stream (`on_ctx_stream`), so collectives and kernels are ordered by the stream itself and
nothing is matched before the broadcast / all_gather that produces it has landed.  The torch
functions `fold_partials` / `accept` remain as the CPU rehearsal of the same rule (gloo tests) and as
the checker of the fold kernel.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import contextlib

import torch
import torch.distributed as dist

NONE = 0xFFFF


@contextlib.contextmanager
def on_ctx_stream(ctx, device=None):
    """Makes the context's HIP stream torch's current stream: torch kernels, RCCL collectives and the
    library's own launches are then ordered by ONE stream (a collective enqueued here waits for what the
    stream holds, and what is enqueued after it waits for the collective)."""
    if ctx is None or not torch.cuda.is_available():
        yield None
        return
    ext = torch.cuda.ExternalStream(ctx.stream(), device=device)
    with torch.cuda.stream(ext):
        yield ext


def shard_cameras(n_cameras: int, world: int, rank: int) -> List[int]:
    """Camera ids (numbered from 1: camera_id 0 is rejected by the shim,
    orbslam3_mono_networked.cc:479,528) handled by `rank`; round-robin, one per GPU first."""
    return [c + 1 for c in range(n_cameras) if c % world == rank]


def slab(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [begin, end) slab of a database of n_items rows for `rank`."""
    per = (n_items + world - 1) // world
    b = min(rank * per, n_items)
    return b, min(b + per, n_items)


def _collective_device(t: torch.Tensor) -> torch.Tensor:
    """gloo moves CPU tensors; nccl (RCCL) moves device tensors in place."""
    return t.cpu() if dist.get_backend() == "gloo" else t


def all_gather_fixed(t: torch.Tensor) -> List[torch.Tensor]:
    world = dist.get_world_size()
    src = _collective_device(t.contiguous())
    out = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(out, src)
    return [o.to(t.device) for o in out]


def fold_partials(d1s: Sequence[torch.Tensor], j1s: Sequence[torch.Tensor], d2s: Sequence[torch.Tensor]):
    """Element-wise merge of per-slab (d1, j1, d2), slabs in ascending global-index order."""
    d1, j1, d2 = d1s[0].clone().long(), j1s[0].clone().long(), d2s[0].clone().long()
    for e1, ej, e2 in zip(d1s[1:], j1s[1:], d2s[1:]):
        e1, ej, e2 = e1.long(), ej.long(), e2.long()
        better = e1 < d1
        d2 = torch.where(better, torch.minimum(d1, e2), torch.minimum(d2, e1))
        j1 = torch.where(better, ej, j1)
        d1 = torch.where(better, e1, d1)
    return d1, j1, d2


def accept(d1: torch.Tensor, j1: torch.Tensor, d2: torch.Tensor, th: int = 50, ratio_num: int = 9,
           ratio_den: int = 10) -> torch.Tensor:
    ok = (j1 >= 0) & (d1 <= th) & (d1 * ratio_den < d2 * ratio_num)
    return torch.where(ok, j1, torch.full_like(j1, -1))


LocalMatch = Callable[[torch.Tensor, torch.Tensor], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]


def hip_local_match(ctx) -> LocalMatch:
    """Raw (th < 0) ss_match_device on this rank's GPU: (query u8[nq,32], train u8[nt,32]) device
    tensors -> (j1 int32, d1, d2 as int32) device tensors.  The inputs may have been produced on torch's
    current stream (a collective's output): the context's stream is ordered after it first."""
    def run(q: torch.Tensor, t: torch.Tensor):
        nq, nt = q.shape[0], t.shape[0]
        idx = torch.empty(nq, dtype=torch.int32, device=q.device)
        d1 = torch.empty(nq, dtype=torch.int16, device=q.device)
        d2 = torch.empty(nq, dtype=torch.int16, device=q.device)
        ctx.wait_stream(torch.cuda.current_stream(q.device).cuda_stream)
        ctx.match_device(q.data_ptr(), nq, t.data_ptr() if nt else 0, nt, idx.data_ptr(), d1.data_ptr(),
                         d2.data_ptr(), th=-1)
        ctx.synchronize()
        return idx, d1.to(torch.int32) & 0xFFFF, d2.to(torch.int32) & 0xFFFF
    run.ctx = ctx
    return run


def expand_database(ctx, db_slab: torch.Tensor) -> torch.Tensor:
    """A rank's database slab (u8 [n, 32]) in the matrix-core matcher's operand format (u8 [n rounded up to 32, 128]),
    prepared ONCE: the per-query match then expands nothing (ss_expand_descriptors_device)."""
    n = db_slab.shape[0]
    out = torch.empty((ctx.expanded_bytes(n) // 128, 128), dtype=torch.uint8, device=db_slab.device)
    with on_ctx_stream(ctx, db_slab.device):
        ctx.expand_descriptors_device(db_slab.data_ptr(), n, out.data_ptr())
    ctx.synchronize()
    return out


def _local_partial(ctx, query, db_slab, slab_begin, part, db_expanded, n_db):
    nq, dev = query.shape[0], query.device
    if db_expanded is not None:
        # the slab was expanded once (expand_database); the query is expanded per call (nq x 128 B)
        nt = db_slab.shape[0] if n_db is None else n_db
        qx = torch.empty((ctx.expanded_bytes(nq) // 128, 128), dtype=torch.uint8, device=dev)
        ctx.expand_descriptors_device(query.data_ptr(), nq, qx.data_ptr())
        ctx.match_partial_expanded_device(qx.data_ptr(), nq, db_expanded.data_ptr() if nt else 0, nt, slab_begin, part.data_ptr())
    else:
        nt = db_slab.shape[0]
        ctx.match_partial_device(query.data_ptr(), nq, db_slab.data_ptr() if nt else 0, nt, slab_begin, part.data_ptr())


def loop_closure_query_native(ctx, xchg, query: torch.Tensor, db_slab: torch.Tensor, slab_begin: int, th: int = 50,
                              ratio_num: int = 9, ratio_den: int = 10, src: int = 0, out=None, db_expanded=None, n_db=None):
    """Config 5 with the C ABI's own exchange (binding.Exchange = ss_xchg_*): ss_xchg_broadcast of the query ->
    ss_match_partial*_device -> ss_xchg_allgather of world x nq x 8 B -> ss_match_fold_strided_device on the gathered
    records where they landed.  No torch.distributed, no RCCL; torch only holds the buffers.  `query` is overwritten with
    rank `src`'s on every other rank (in place, like ncclBroadcast)."""
    nq, dev = query.shape[0], query.device
    with on_ctx_stream(ctx, dev):  # the two scratch tensors below are allocated and freed in the context's stream order
        xchg.broadcast(ctx, src, query.data_ptr(), nq * 32)
        part = torch.empty(nq, dtype=torch.int64, device=dev)
        _local_partial(ctx, query, db_slab, slab_begin, part, db_expanded, n_db)
        base, stride = xchg.allgather(ctx, [(part.data_ptr(), nq * 8)])
        if out is None:
            out = (torch.empty(nq, dtype=torch.int32, device=dev), torch.empty(nq, dtype=torch.int16, device=dev),
                   torch.empty(nq, dtype=torch.int16, device=dev))
        ctx.match_fold_strided_device(base, xchg.world, stride, nq, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                      th=th, ratio_num=ratio_num, ratio_den=ratio_den)
    return out


def loop_closure_query_device(ctx, query: torch.Tensor, db_slab: torch.Tensor, slab_begin: int, th: int = 50,
                              ratio_num: int = 9, ratio_den: int = 10, src: int = 0, out=None, db_expanded=None, n_db=None,
                              xchg=None):
    """Config 5 on the GPUs, behind the C ABI: broadcast -> ss_match_partial_device -> all_gather of
    world x nq x 8 B -> ss_match_fold_device, all on the context's stream (no host synchronisation inside;
    the caller synchronises when it reads the result).  Returns device tensors (idx i32, d1 i16, d2 i16:
    the 16-bit distances, 0xFFFF = none).  With `xchg` (binding.Exchange) the two collectives are the library's own."""
    if xchg is not None:
        return loop_closure_query_native(ctx, xchg, query, db_slab, slab_begin, th, ratio_num, ratio_den, src, out, db_expanded, n_db)
    world = dist.get_world_size() if dist.is_initialized() else 1
    nq, dev = query.shape[0], query.device
    with on_ctx_stream(ctx, dev):
        if world > 1:
            if dist.get_backend() == "gloo":
                q = query.cpu()
                dist.broadcast(q, src=src)
                query = q.to(dev)
            else:
                dist.broadcast(query, src=src)
        part = torch.empty(nq, dtype=torch.int64, device=dev)  # nq x ss_match_part (8 B)
        _local_partial(ctx, query, db_slab, slab_begin, part, db_expanded, n_db)
        if world > 1:
            if dist.get_backend() == "gloo":
                ctx.synchronize()
                src_t = part.cpu()
                outs = [torch.empty_like(src_t) for _ in range(world)]
                dist.all_gather(outs, src_t)
                parts = torch.stack(outs).to(dev)
            else:
                parts = torch.empty((world, nq), dtype=torch.int64, device=dev)
                dist.all_gather_into_tensor(parts, part)
        else:
            parts = part.view(1, nq)
        if out is None:
            out = (torch.empty(nq, dtype=torch.int32, device=dev), torch.empty(nq, dtype=torch.int16, device=dev),
                   torch.empty(nq, dtype=torch.int16, device=dev))
        ctx.match_fold_device(parts.data_ptr(), world, nq, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                              th=th, ratio_num=ratio_num, ratio_den=ratio_den)
    return out


def loop_closure_query(query: torch.Tensor, db_slab: torch.Tensor, slab_begin: int, local_match: LocalMatch,
                       th: int = 50, ratio_num: int = 9, ratio_den: int = 10, src: int = 0):
    """Config 5.  `query` (u8 [nq,32]) is taken from rank `src` and broadcast; `db_slab` is this
    rank's contiguous part of the keyframe-descriptor database, starting at global row
    `slab_begin`.  Every rank returns the same (idx, d1, d2) over the WHOLE database."""
    q = _collective_device(query.contiguous())
    dist.broadcast(q, src=src)
    q = q.to(query.device)
    j1, d1, d2 = local_match(q, db_slab)
    j1 = torch.where(j1 >= 0, j1.long() + slab_begin, j1.long())
    part = torch.stack([d1.long(), j1, d2.long()])  # 3 x nq
    parts = all_gather_fixed(part)
    fd1, fj1, fd2 = fold_partials([p[0] for p in parts], [p[1] for p in parts], [p[2] for p in parts])
    return accept(fd1, fj1, fd2, th, ratio_num, ratio_den).to(torch.int32), fd1, fd2


def stereo_exchange(desc: torch.Tensor, n_kp: int, local_match: LocalMatch, th: int = 50, ratio_num: int = 9,
                    ratio_den: int = 10):
    """Config 4 (world size 2).  `desc` is this rank's fixed-size descriptor block
    (u8 [kp_capacity, 32], rows >= n_kp unused).  Returns matches of this eye's keypoints
    against the peer eye's: (idx into the peer's rows, d1, d2, peer_n_kp)."""
    assert dist.get_world_size() == 2
    rank = dist.get_rank()
    with on_ctx_stream(getattr(local_match, "ctx", None), desc.device if desc.is_cuda else None):
        blocks = all_gather_fixed(desc)
        counts = all_gather_fixed(torch.tensor([n_kp], dtype=torch.int64, device=desc.device))
        peer = 1 - rank
        peer_n = int(counts[peer].item())
    j1, d1, d2 = local_match(desc[:n_kp], blocks[peer][:peer_n])
    idx = accept(d1.long(), j1.long(), d2.long(), th, ratio_num, ratio_den).to(torch.int32)
    return idx, d1, d2, peer_n
