import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/send-slam_amd")
import numpy as np, torch
from send_slam_amd import binding, synth
frames = np.stack([synth.frame(i, 1280, 720) for i in range(2)])
d = torch.from_numpy(frames).to("cuda:0")
with binding.OrbContext(0, n_features=2000, max_batch=2) as ctx:
    ctx.extract_batch_device(d.data_ptr(), 2, 1280, 720)
    ctx.synchronize()
