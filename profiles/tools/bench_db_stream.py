"""Hamming-match kernel in the database-streaming regime (BASELINE.json config 5, one GPU's shard
scaled to a full 640 MB database): a few query descriptors against 20 M keyframe descriptors,
database resident in HBM.  Reports the kernel's HIP-event time and achieved GB/s against the
8 TB/s HBM peak.  Run on the GPU box:  python profiles/tools/bench_db_stream.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "send-slam_amd"))
import torch  # noqa: E402
from send_slam_amd import binding  # noqa: E402

dev = torch.device("cuda:0")
nt = 20_000_000  # 10 000 keyframes x 2000 descriptors = 640 MB
db = torch.randint(0, 256, (nt, 32), dtype=torch.uint8, device=dev)
out = []
with binding.OrbContext(0) as ctx:
    for nq in (1, 2, 4, 8, 64, 2000):
        q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device=dev)
        idx = torch.empty(nq, dtype=torch.int32, device=dev)
        d1 = torch.empty(nq, dtype=torch.int16, device=dev)
        d2 = torch.empty(nq, dtype=torch.int16, device=dev)
        for _ in range(2):
            ctx.match_device(q.data_ptr(), nq, db.data_ptr(), nt, idx.data_ptr(), d1.data_ptr(), d2.data_ptr(), th=-1)
        ctx.synchronize()
        ctx.profile(True)
        ctx.profile_reset()
        reps = 5 if nq <= 64 else 2
        for _ in range(reps):
            ctx.match_device(q.data_ptr(), nq, db.data_ptr(), nt, idx.data_ptr(), d1.data_ptr(), d2.data_ptr(), th=-1)
        ctx.synchronize()
        st = [s for s in ctx.stats() if s["launches"]]
        ctx.profile(False)
        ms = sum(s["total_ms"] for s in st) / reps
        nbytes = nt * 32 + nq * 40
        rec = {"n_query": nq, "n_train": nt, "kernel": st[0]["name"], "ms": round(ms, 4), "algorithmic_bytes": nbytes,
               "achieved_GBps": round(nbytes / (ms * 1e-3) / 1e9, 1), "frac_of_8TBps": round(nbytes / (ms * 1e-3) / 8e12, 4),
               "pairs_per_s": round(nq * nt / (ms * 1e-3) / 1e12, 3)}
        out.append(rec)
        print(json.dumps(rec))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "db_stream.json"), "w"), indent=1)
