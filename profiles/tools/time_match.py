import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "send-slam_amd"))
from send_slam_amd import binding, synth
if os.environ.get("SENDSLAM_LIB"): binding.LIB_PATH = os.environ["SENDSLAM_LIB"]
B, n = 64, 2000
w, h = 1280, 720
ctx = binding.OrbContext(0, n_features=n, max_batch=B)
sc = [synth.scene(1000 + i, w, h) for i in range(8)]
frames = np.stack([synth.frame_from_scene(sc[i % 8], 1000 + i % 8, w, h, i // 8) for i in range(B)])
d = torch.from_numpy(frames).cuda()
ctx.extract_batch_device(d.data_ptr(), B, w, h)
kcap = 4096
idx = torch.empty(B * kcap, dtype=torch.int32, device="cuda"); d1 = torch.empty(B * kcap, dtype=torch.int16, device="cuda"); d2 = torch.empty_like(d1)
for _ in range(3): ctx.match_batch_device(0, idx.data_ptr(), d1.data_ptr(), d2.data_ptr())
ctx.synchronize(); ctx.profile(True)
try: ctx.profile_reset()
except Exception: pass
for _ in range(20): ctx.match_batch_device(0, idx.data_ptr(), d1.data_ptr(), d2.data_ptr())
ctx.synchronize()
for s in ctx.stats():
    if s["launches"] and "match" in s["name"]: print(os.path.basename(os.environ.get("SENDSLAM_LIB", "default")), s["name"], round(s["total_ms"] / 20, 4), s["launches"])
