"""The C ABI's own exchange (ss_xchg_*, csrc/ss_xchg.hip): SURVEY.md section 8(e)'s one-hop direct-write all-gather over
peer-mapped device memory, here between two PROCESSES ON ONE CARD (hipIpc works same-device; the driver's 8-GPU node is
where it runs over xGMI).  Configs 4 and 5 through it, no torch.distributed, results against the oracle; 120 messages of
changing sizes with every byte checked under uneven load."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import multi_worker
from send_slam_amd import binding


def test_exchange_needs_a_device_and_sane_arguments():
    lib = binding.load()
    import ctypes as C
    h = C.c_void_p()
    assert lib.ss_xchg_create(0, 2, 2, 1024, b"/tmp/x", 100, C.byref(h)) == binding.SS_ERR_INVALID_ARG  # rank >= world
    assert lib.ss_xchg_create(0, 0, 2, 1024, None, 100, C.byref(h)) == binding.SS_ERR_INVALID_ARG       # no rendezvous
    assert b"rendezvous" in lib.ss_xchg_last_error(None)
    if not torch.cuda.is_available():
        assert lib.ss_xchg_create(0, 0, 1, 1024, None, 100, C.byref(h)) == binding.SS_ERR_NO_DEVICE
        assert b"no CPU path" in lib.ss_xchg_last_error(None)
    assert lib.ss_xchg_status(None) == binding.SS_ERR_INVALID_ARG and lib.ss_xchg_destroy(None) == binding.SS_ERR_INVALID_ARG


@pytest.mark.gpu
def test_world1_exchange_returns_own_block():
    dev = torch.device("cuda:0")
    with binding.OrbContext(0) as ctx, binding.Exchange(0, 0, 1, 4096, "") as x:
        a = torch.arange(0, 1000, device=dev).to(torch.uint8)
        b = torch.full((36,), 7, dtype=torch.uint8, device=dev)
        base, stride = x.allgather(ctx, [(a.data_ptr(), 1000), (b.data_ptr(), 36)])
        assert stride == 1008 + 48
        got = multi_worker.dev_view(torch, base, stride, dev).clone()
        ctx.synchronize()
        x.status()
        got = got.cpu().numpy()
        assert np.array_equal(got[:1000], a.cpu().numpy()) and (got[1008:1008 + 36] == 7).all()
        with pytest.raises(binding.OrbError):
            x.allgather(ctx, [(a.data_ptr(), 5000)])  # larger than max_bytes


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_processes_on_one_card_configs_4_and_5(tmp_path, oracle, world):
    """world = 4: four slabs / four peers per message -- the flag and slot indexing beyond two ranks, which is what an 8-GPU node
    runs and no 2-rank test touches (four processes share the one card; IPC mappings work on the same device)."""
    path = str(tmp_path / "xchg.sock")
    mp.spawn(multi_worker.run_xchg, args=(world, path, str(tmp_path)), nprocs=world, join=True)
    for k, seed in enumerate((5, 6, 7)):
        q, db = multi_worker.make_db(seed, 4001, 150)
        want = oracle.match(q, db, th=256, ratio_num=10)
        for r in range(world):
            z = np.load(os.path.join(str(tmp_path), f"xlc{k}_{r}.npz"))
            assert np.array_equal(z["query"], q), "the broadcast did not deliver rank 0's query"
            assert np.array_equal(z["idx"], want[0]) and np.array_equal(z["d1"], want[1]) and np.array_equal(z["d2"], want[2]), (k, r)
    rng = np.random.default_rng(100)
    kcap, B = 192, 3
    eyes = [rng.integers(0, 256, size=(B, kcap, 32), dtype=np.uint8) for _ in range(2)]
    counts = [np.array([150, 171, 0], np.int32), np.array([171, 128, 192], np.int32)]
    eyes[1][:, :100] = eyes[0][:, :100]
    eyes[1][:, :100, 4] ^= 0x0F
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"xst_{r}.npz"))
        me = r % 2  # rank r holds eye r % 2 and matches it against rank r + 1's
        for b in range(B):
            nq, nt = counts[me][b], counts[1 - me][b]
            w = oracle.match(eyes[me][b, :nq], eyes[1 - me][b, :nt])
            assert np.array_equal(z["idx"][b, :nq], w[0]) and np.array_equal(z["d1"][b, :nq], w[1]) and np.array_equal(z["d2"][b, :nq], w[2]), (r, b)
            assert (z["idx"][b, nq:] == -1).all()
        assert int(np.load(os.path.join(str(tmp_path), f"xmsg_{r}.npz"))["bad"]) == 0
