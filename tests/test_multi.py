"""Multi-GPU shapes: the sharding, exchange and fold logic of
send_slam_amd/multi.py under a world_size-2 gloo group.  CPU run: each rank's local matcher
is the oracle (test infrastructure), so what is tested is the collective + fold.  GPU run
(one card, two ranks): the local matcher is ss_match_device in raw mode."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import multi_worker
from send_slam_amd import multi


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def check_device_path_outputs(out_dir, oracle):
    for k, seed in enumerate((5, 6, 7)):
        q, db = multi_worker.make_db(seed, 4001, 150)
        want = oracle.match(q, db, th=256, ratio_num=10)
        for r in range(2):
            z = np.load(os.path.join(out_dir, f"lcd{k}_{r}.npz"))
            assert np.array_equal(z["idx"], want[0]) and np.array_equal(z["d1"], want[1]) and np.array_equal(z["d2"], want[2]), (k, r)


def check_outputs(out_dir, oracle):
    q, db = multi_worker.make_db(5, 4001, 150)
    want = oracle.match(q, db, th=256, ratio_num=10)
    for r in range(2):
        z = np.load(os.path.join(out_dir, f"lc_{r}.npz"))
        assert np.array_equal(z["idx"], want[0]) and np.array_equal(z["d1"], want[1]) and np.array_equal(z["d2"], want[2])
    assert want[0][0] == -1 and want[1][0] == 0 and want[2][0] == 0       # duplicate across slabs: ratio rejects
    assert want[0][1] == 2000 + 9 and want[0][2] == 7 and want[2][2] == 2  # bests and runner-up in different slabs
    rng = np.random.default_rng(100)
    eyes = [rng.integers(0, 256, size=(192, 32), dtype=np.uint8) for _ in range(2)]
    counts = [150, 171]
    eyes[1][:100] = eyes[0][:100]
    eyes[1][:100, 4] ^= 0x0F
    for r in range(2):
        z = np.load(os.path.join(out_dir, f"st_{r}.npz"))
        w = oracle.match(eyes[r][:counts[r]], eyes[1 - r][:counts[1 - r]])
        assert int(z["peer_n"]) == counts[1 - r]
        assert np.array_equal(z["idx"], w[0]) and np.array_equal(z["d1"], w[1]) and np.array_equal(z["d2"], w[2])
        assert (z["idx"][:100] == np.arange(100)).all()


def test_shard_cameras_and_slabs():
    # config 3: 8 cameras over 8 GPUs -> one each, ids from 1 (camera_id 0 is rejected by the shim)
    assert [multi.shard_cameras(8, 8, r) for r in range(8)] == [[i + 1] for i in range(8)]
    assert multi.shard_cameras(10, 4, 1) == [2, 6, 10] and multi.shard_cameras(2, 4, 3) == []
    # config 5: 10 000 keyframes x 2000 descriptors over 8 GPUs -> 1250 keyframes (80 MB) per GPU
    n = 10000 * 2000
    slabs = [multi.slab(n, 8, r) for r in range(8)]
    assert slabs[0] == (0, 2500000) and slabs[7] == (17500000, n) and all(e - b == 2500000 for b, e in slabs)
    assert (2500000 * 32) == 80_000_000
    assert multi.slab(5, 4, 3) == (5, 5) and multi.slab(5, 4, 2) == (4, 5)


def test_fold_rule_matches_sequential_scan():
    rng = np.random.default_rng(3)
    for _ in range(200):
        d = rng.integers(0, 6, size=12)  # many ties
        cuts = sorted(rng.choice(np.arange(1, 12), size=2, replace=False))
        parts = np.split(np.arange(12), cuts)
        d1s, j1s, d2s = [], [], []
        for p in parts:
            order = sorted(p, key=lambda j: (d[j], j))
            d1s.append(torch.tensor([d[order[0]]]))
            j1s.append(torch.tensor([order[0]]))
            d2s.append(torch.tensor([d[order[1]] if len(order) > 1 else multi.NONE]))
        fd1, fj1, fd2 = multi.fold_partials(d1s, j1s, d2s)
        order = sorted(range(12), key=lambda j: (d[j], j))
        assert (int(fd1), int(fj1), int(fd2)) == (d[order[0]], order[0], d[order[1]])


def test_world2_gloo_cpu(tmp_path, oracle):
    port = free_port()
    mp.spawn(multi_worker.run, args=(2, port, False, str(tmp_path)), nprocs=2, join=True)
    check_outputs(str(tmp_path), oracle)


@pytest.mark.gpu
def test_world2_gloo_two_ranks_one_gpu(tmp_path, oracle):
    port = free_port()
    mp.spawn(multi_worker.run, args=(2, port, True, str(tmp_path)), nprocs=2, join=True)
    check_outputs(str(tmp_path), oracle)
    check_device_path_outputs(str(tmp_path), oracle)


@pytest.mark.gpu
def test_fold_kernel_equals_torch_fold(oracle):
    """ss_match_fold_device (one launch of k_match_merge's rule) against multi.fold_partials + accept on random
    records with many ties and empty shards."""
    from send_slam_amd import binding
    rng = np.random.default_rng(8)
    dev = torch.device("cuda:0")
    with binding.OrbContext(0) as ctx:
        for n_parts, nq in ((2, 100), (8, 2000), (5, 1)):
            d1 = rng.integers(0, 6, size=(n_parts, nq)).astype(np.uint16)
            d2 = (d1 + rng.integers(0, 3, size=(n_parts, nq))).astype(np.uint16)
            row = (np.arange(n_parts)[:, None] * 1000 + rng.integers(0, 1000, size=(n_parts, nq))).astype(np.int32)
            empty = rng.random((n_parts, nq)) < 0.2
            d1[empty], d2[empty], row[empty] = 0xFFFF, 0xFFFF, -1
            rec = np.zeros((n_parts, nq), np.dtype([("d1", "<u2"), ("d2", "<u2"), ("row", "<i4")]))
            rec["d1"], rec["d2"], rec["row"] = d1, d2, row
            parts = torch.from_numpy(rec.view(np.int64).reshape(n_parts, nq)).to(dev)
            idx = torch.empty(nq, dtype=torch.int32, device=dev)
            o1 = torch.empty(nq, dtype=torch.int16, device=dev)
            o2 = torch.empty(nq, dtype=torch.int16, device=dev)
            ctx.match_fold_device(parts.data_ptr(), n_parts, nq, idx.data_ptr(), o1.data_ptr(), o2.data_ptr(), th=4, ratio_num=9)
            ctx.synchronize()
            t = lambda a: [torch.from_numpy(a[p].astype(np.int64)) for p in range(n_parts)]  # noqa: E731
            fd1, fj1, fd2 = multi.fold_partials(t(d1), t(row), t(d2))
            want_idx = multi.accept(fd1, fj1, fd2, th=4, ratio_num=9, ratio_den=10)
            assert np.array_equal(idx.cpu().numpy(), want_idx.numpy().astype(np.int32))
            assert np.array_equal(o1.cpu().numpy().view(np.uint16), fd1.numpy().astype(np.uint16))
            assert np.array_equal(o2.cpu().numpy().view(np.uint16), fd2.numpy().astype(np.uint16))


@pytest.mark.gpu
def test_raw_mode_and_fold_over_slabs_single_process(oracle):
    from send_slam_amd import binding
    q, db = multi_worker.make_db(9, 30001, 333)
    dev = torch.device("cuda:0")
    with binding.OrbContext(0) as ctx:
        local = multi.hip_local_match(ctx)
        tq = torch.from_numpy(q).to(dev)
        parts = []
        for r in range(3):
            b, e = multi.slab(len(db), 3, r)
            j1, d1, d2 = local(tq, torch.from_numpy(db[b:e]).to(dev))
            raw = oracle.match(q, db[b:e], th=-1)
            assert np.array_equal(j1.cpu().numpy(), raw[0]) and np.array_equal(d1.cpu().numpy(), raw[1])
            parts.append((d1.long(), torch.where(j1 >= 0, j1.long() + b, j1.long()), d2.long()))
        fd1, fj1, fd2 = multi.fold_partials([p[0] for p in parts], [p[1] for p in parts], [p[2] for p in parts])
        idx = multi.accept(fd1, fj1, fd2).cpu().numpy()
    want = oracle.match(q, db)
    assert np.array_equal(idx, want[0]) and np.array_equal(fd1.cpu().numpy(), want[1]) and np.array_equal(fd2.cpu().numpy(), want[2])
