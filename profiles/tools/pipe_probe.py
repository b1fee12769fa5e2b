"""Where a pipe batch's time goes: host time inside submit / wait, and the rate at several depths.
usage: python profiles/tools/pipe_probe.py [batch] [depth]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "send-slam_amd"))
import numpy as np  # noqa: E402

from send_slam_amd import binding, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w, h, nf = 1280, 720, 2000
sc = synth.scene(0, w, h)
frames = np.stack([synth.frame_from_scene(sc, 0, w, h, t) for t in range(B)])
for depth in ([int(sys.argv[2])] if len(sys.argv) > 2 else [2, 3, 4, 6]):
    for match_mode in (0, -1):
        with binding.Pipe(0, w, h, batch=B, depth=depth, match_mode=match_mode, n_features=nf) as pipe:
            slots = []
            for s in range(depth):
                sl = pipe.acquire()
                sl[1][:B, :, :w] = frames
                slots.append(sl[0])
            for s in slots:
                pipe.release(s)
            t_sub, t_wait = [], []
            n = 120
            t0 = time.perf_counter()
            for i in range(n):
                if pipe.in_flight() == depth:
                    t1 = time.perf_counter()
                    r = pipe.wait()
                    t_wait.append(time.perf_counter() - t1)
                    pipe.release(r["slot"])
                t1 = time.perf_counter()
                pipe.submit(pipe.acquire()[0], B)
                t_sub.append(time.perf_counter() - t1)
            while pipe.in_flight():
                r = pipe.wait()
                pipe.release(r["slot"])
            el = time.perf_counter() - t0
            print(f"depth {depth} match {match_mode}: {n * B / el:9.0f} frames/s  {el / n * 1e3:.3f} ms/batch  submit {np.median(t_sub) * 1e3:.3f} ms (max {max(t_sub) * 1e3:.3f})  "
                  f"wait {np.median(t_wait) * 1e3:.3f} ms", flush=True)
