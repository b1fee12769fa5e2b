"""Worker entry points for the world_size-2 tests (spawned processes; test infrastructure)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "send-slam_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def oracle_local_match():
    """CPU stand-in for the per-rank matcher in gloo rehearsals: the ORACLE in raw mode.  Test
    infrastructure only: it checks the collective / fold logic, never the product kernel."""
    import torch
    from oracle import orb_oracle as O

    def run(q, t):
        idx, d1, d2 = O.match(q.cpu().numpy(), t.cpu().numpy().reshape(-1, 32), th=-1)
        return (torch.from_numpy(idx.astype(np.int32)), torch.from_numpy(d1.astype(np.int32)),
                torch.from_numpy(d2.astype(np.int32)))
    return run


def make_db(seed, n_db, nq):
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 256, size=(nq, 32), dtype=np.uint8)
    db = rng.integers(0, 256, size=(n_db, 32), dtype=np.uint8)
    half = n_db // 2
    db[3] = q[0]; db[half + 5] = q[0]          # exact duplicate in both slabs: lowest index wins, d2 = 0
    db[half + 9] = q[1]                        # best lives in the second slab
    db[7] = q[2]; db[7, 0] ^= 1                # distance-1 best in the first slab
    db[half + 1] = q[2]; db[half + 1, 5] ^= 3  # distance-2 runner-up in the second slab
    return q, db


def run(rank, world, port, use_gpu, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    from send_slam_amd import multi
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0") if use_gpu else torch.device("cpu")
        if use_gpu:
            from send_slam_amd import binding
            ctx = binding.OrbContext(0)
            local = multi.hip_local_match(ctx)
        else:
            local = oracle_local_match()
        # config 5: loop closure over a partitioned database
        q, db = make_db(5, 4001, 150)
        b, e = multi.slab(len(db), world, rank)
        query = torch.from_numpy(q if rank == 0 else np.zeros_like(q)).to(dev)
        idx, d1, d2 = multi.loop_closure_query(query, torch.from_numpy(db[b:e]).to(dev), b, local, th=256, ratio_num=10)
        np.savez(os.path.join(out_dir, f"lc_{rank}.npz"), idx=idx.cpu().numpy(), d1=d1.cpu().numpy(), d2=d2.cpu().numpy())
        # config 4: stereo exchange of fixed-size descriptor blocks
        rng = np.random.default_rng(100)
        kcap = 192
        eyes = [rng.integers(0, 256, size=(kcap, 32), dtype=np.uint8) for _ in range(2)]
        counts = [150, 171]
        eyes[1][:100] = eyes[0][:100]
        eyes[1][:100, 4] ^= 0x0F  # the right eye sees the left eye's points 4 bits away
        sidx, sd1, sd2, peer_n = multi.stereo_exchange(torch.from_numpy(eyes[rank]).to(dev), counts[rank], local)
        np.savez(os.path.join(out_dir, f"st_{rank}.npz"), idx=sidx.cpu().numpy(), d1=sd1.cpu().numpy(), d2=sd2.cpu().numpy(),
                 peer_n=peer_n)
        if use_gpu:
            # the same query through the C-ABI path (partial -> all_gather -> fold kernel), then a DIFFERENT query per
            # call, each taken from a device tensor on rank 0: a stale query or a stale gathered block would show
            for k, seed in enumerate((5, 6, 7)):
                qk, dbk = make_db(seed, 4001, 150)
                query = torch.from_numpy(qk if rank == 0 else np.zeros_like(qk)).to(dev)
                slab_t = torch.from_numpy(dbk[b:e]).to(dev)
                kw = {"db_expanded": multi.expand_database(ctx, slab_t)} if k == 2 else {}  # the last one on the expanded slab
                idx, d1, d2 = multi.loop_closure_query_device(ctx, query, slab_t, b, th=256, ratio_num=10, **kw)
                ctx.synchronize()
                np.savez(os.path.join(out_dir, f"lcd{k}_{rank}.npz"), idx=idx.cpu().numpy(), d1=d1.cpu().numpy().view(np.uint16),
                         d2=d2.cpu().numpy().view(np.uint16))
            ctx.close()
    finally:
        dist.destroy_process_group()


def dev_view(torch, ptr, nbytes, dev):
    """uint8 torch view of `nbytes` of device memory at `ptr` (no copy)"""
    class _Wrap:
        __cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
    return torch.as_tensor(_Wrap(), device=dev)


def run_xchg(rank, world, path, out_dir, one_device=True):
    """The C ABI's own exchange (ss_xchg_*) between `world` processes: configs 4 and 5 without torch.distributed."""
    os.environ.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    from send_slam_amd import binding, multi
    d = 0 if one_device else rank
    torch.cuda.set_device(d)
    dev = torch.device("cuda", d)
    ctx = binding.OrbContext(d)
    kcap, B = 192, 3
    xchg = binding.Exchange(d, rank, world, max(B * kcap * 32 + 64, 150 * 32), path, timeout_ms=20000)
    try:
        # config 5: loop closure, three different queries, the last on the expanded slab
        for k, seed in enumerate((5, 6, 7)):
            qk, dbk = make_db(seed, 4001, 150)
            b, e = multi.slab(len(dbk), world, rank)
            query = torch.from_numpy(qk if rank == 0 else np.zeros_like(qk)).to(dev)
            slab_t = torch.from_numpy(dbk[b:e]).to(dev)
            kw = {"db_expanded": multi.expand_database(ctx, slab_t)} if k == 2 else {}
            idx, d1, d2 = multi.loop_closure_query_device(ctx, query, slab_t, b, th=256, ratio_num=10, xchg=xchg, **kw)
            ctx.synchronize()
            xchg.status()
            np.savez(os.path.join(out_dir, f"xlc{k}_{rank}.npz"), idx=idx.cpu().numpy(), d1=d1.cpu().numpy().view(np.uint16),
                     d2=d2.cpu().numpy().view(np.uint16), query=query.cpu().numpy())
        # config 4: B frames per eye, fixed-size blocks + counts in ONE message of two segments, cross-eye match
        rng = np.random.default_rng(100)
        eyes = [rng.integers(0, 256, size=(B, kcap, 32), dtype=np.uint8) for _ in range(2)]
        counts = [np.array([150, 171, 0], np.int32), np.array([171, 128, 192], np.int32)]
        eyes[1][:, :100] = eyes[0][:, :100]
        eyes[1][:, :100, 4] ^= 0x0F
        me = rank % 2
        own = torch.from_numpy(eyes[me]).to(dev)
        own_n = torch.from_numpy(counts[me]).to(dev)
        o_idx = torch.empty((B, kcap), dtype=torch.int32, device=dev)
        o_d1 = torch.empty((B, kcap), dtype=torch.int16, device=dev)
        o_d2 = torch.empty((B, kcap), dtype=torch.int16, device=dev)
        blk = B * kcap * 32
        base, stride = xchg.allgather(ctx, [(own.data_ptr(), blk), (own_n.data_ptr(), 4 * B)])
        peer = (rank + 1) % world
        ctx.match_pairs_device(own.data_ptr(), own_n.data_ptr(), base + peer * stride, base + peer * stride + blk, B, kcap,
                               o_idx.data_ptr(), o_d1.data_ptr(), o_d2.data_ptr())
        ctx.synchronize()
        xchg.status()
        np.savez(os.path.join(out_dir, f"xst_{rank}.npz"), idx=o_idx.cpu().numpy(), d1=o_d1.cpu().numpy().view(np.uint16),
                 d2=o_d2.cpu().numpy().view(np.uint16))
        # many messages of changing sizes and contents, every byte checked: a stale parity buffer, a flag that overtakes
        # its data or a lost store would show.  Uneven load: the odd ranks run extra work between messages.
        bad = 0
        ext = torch.cuda.ExternalStream(ctx.stream(), device=dev)
        busy = torch.empty(8 << 20, dtype=torch.float32, device=dev)
        for it in range(120):
            n = 16 * (1 + (it * 37) % 1000) + (it % 3) * 5  # also sizes that are not multiples of 16
            with torch.cuda.stream(ext):
                if rank % 2 == 1 and it % 4 == 0:
                    busy.mul_(1.0001)
                msg = torch.full((n,), (it * 7 + rank * 13) & 0xFF, dtype=torch.uint8, device=dev)
                msg[::97] = torch.arange(0, (n + 96) // 97, device=dev).to(torch.uint8)
                base, stride = xchg.allgather(ctx, [(msg.data_ptr(), n)])
                got = dev_view(torch, base, stride * world, dev).clone()
            ctx.synchronize()
            got = got.cpu().numpy().reshape(world, stride)
            for r in range(world):
                want = np.full(n, (it * 7 + r * 13) & 0xFF, np.uint8)
                want[::97] = np.arange(0, (n + 96) // 97).astype(np.uint8)
                bad += int(not np.array_equal(got[r, :n], want))
            if it % 5 == 0:  # a broadcast in between, from a changing root
                root = (it // 5) % world
                with torch.cuda.stream(ext):
                    buf = torch.full((n,), (it + 1) & 0xFF if rank == root else 0, dtype=torch.uint8, device=dev)
                    xchg.broadcast(ctx, root, buf.data_ptr(), n)
                ctx.synchronize()
                bad += int(not bool((buf == ((it + 1) & 0xFF)).all()))
        xchg.status()
        np.savez(os.path.join(out_dir, f"xmsg_{rank}.npz"), bad=bad)
    finally:
        xchg.close()
        ctx.close()
