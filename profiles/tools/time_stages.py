"""Per-stage HIP-event times of one context alone on the chip (no parity check): for kernel experiments.
usage: python profiles/tools/time_stages.py [batch] [reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "send-slam_amd"))
import torch  # noqa: E402
from send_slam_amd import binding, synth  # noqa: E402

if os.environ.get("SENDSLAM_LIB"):
    binding.LIB_PATH = os.environ["SENDSLAM_LIB"]

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
w, h, nf = 1280, 720, 2000
sc = [synth.scene(1000 + i, w, h) for i in range(8)]
frames = np.stack([synth.frame_from_scene(sc[i % 8], 1000 + i % 8, w, h, i // 8) for i in range(B)])
d = torch.from_numpy(frames).cuda()
ctx = binding.OrbContext(0, n_features=nf, max_batch=B)
for _ in range(3):
    ctx.extract_batch_device(d.data_ptr(), B, w, h)
ctx.synchronize()
ctx.profile(True)
for _ in range(reps):
    ctx.extract_batch_device(d.data_ptr(), B, w, h)
ctx.synchronize()
tot = 0.0
for s in ctx.stats():
    if s["launches"]:
        per = s["total_ms"] / reps
        tot += per
        print(f"{s['name']:18s} {per:8.4f} ms/batch  ({s['launches'] // reps} launches)")
print(f"{'sum':18s} {tot:8.4f} ms/batch")
