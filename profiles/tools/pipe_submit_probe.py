"""Host time of ss_pipe_submit against the number of frames in the batch (GPU box): is it the enqueue calls (constant) or
something per frame?  args: width height channels"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "send-slam_amd"))
from send_slam_amd import binding, synth
w, h, ch = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (640, 480, 3)
img = synth.frame(7, w, h, 0)
if ch == 3:
    img = np.repeat(img[:, :, None], 3, axis=2)
cam = binding.Camera(type=b"PinHole", fx=500, fy=500, cx=w / 2, cy=h / 2, k1=0, k2=0, p1=0, p2=0, width=w, height=h, fps=30, rgb=1,
                     th_depth=40.0, baseline=0.0, depth_map_factor=1000.0)
for mode in (-1, 1):
    with binding.Pipe(0, w, h, channels=ch, batch=64, depth=3, match_mode=mode, cam=cam, n_features=1250) as pipe:
        for n in (1, 16, 32, 64, 16, 32, 64):
            slot, pix = pipe.acquire()
            for i in range(n):
                pix[i, :, :w * ch] = img.reshape(h, w * ch)
            t0 = time.perf_counter()
            pipe.submit(slot, n, timestamps=[i / 30 for i in range(n)])
            t1 = time.perf_counter()
            r = pipe.wait()
            t2 = time.perf_counter()
            pipe.release(r["slot"])
            print(f"match_mode {mode}: {n} frames: submit {1e3 * (t1 - t0):.3f} ms, until done {1e3 * (t2 - t0):.3f} ms", flush=True)
# back to back: the second and third submission find the GPU busy with the one before
with binding.Pipe(0, w, h, channels=ch, batch=16, depth=3, match_mode=1, cam=cam, n_features=1250) as pipe:
    for rep in range(3):
        slots = []
        ts = []
        for k in range(3):
            slot, pix = pipe.acquire()
            for i in range(16):
                pix[i, :, :w * ch] = img.reshape(h, w * ch)
            t0 = time.perf_counter()
            pipe.submit(slot, 16, timestamps=[i / 30 for i in range(16)])
            ts.append(time.perf_counter() - t0)
        for k in range(3):
            r = pipe.wait()
            pipe.release(r["slot"])
        print("three submissions of 16 frames back to back:", " ".join(f"{1e3 * t:.3f}" for t in ts), "ms", flush=True)
