"""Host time of ss_pipe_submit_frames (gather of pageable frames into the pinned slot) by copy_threads."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "send-slam_amd"))
import numpy as np  # noqa: E402

from send_slam_amd import binding, synth  # noqa: E402

B, w, h, nf = 64, 1280, 720, 2000
sc = synth.scene(0, w, h)
one = np.stack([synth.frame_from_scene(sc, 0, w, h, t) for t in range(B)])
sets = np.stack([np.roll(one, s, axis=0) for s in range(8)])  # 472 MB of pageable frames
print("cpus", len(os.sched_getaffinity(0)))
for threads in (1, 2, 4, 8, 16, 32):
    with binding.Pipe(0, w, h, batch=B, depth=4, match_mode=0, copy_threads=threads, n_features=nf) as pipe:
        t_sub = []
        n = 48
        t0 = time.perf_counter()
        for i in range(n):
            if pipe.in_flight() == 4:
                r = pipe.wait()
                pipe.release(r["slot"])
            t1 = time.perf_counter()
            assert pipe.submit_batch_array(sets[i % 8])
            t_sub.append(time.perf_counter() - t1)
        while pipe.in_flight():
            r = pipe.wait()
            pipe.release(r["slot"])
        el = time.perf_counter() - t0
        med = np.median(t_sub)
        print(f"copy_threads {threads:2d}: {n * B / el:8.0f} frames/s, submit_frames host time median {med * 1e3:.3f} ms = {B * w * h / med / 1e9:.1f} GB/s gather", flush=True)
