"""Where does the front door bench's sequence lose track?  Plays bench_frontdoor's sequence through ss_track frame by frame and
prints state / matches / inliers / map points per frame (GPU box).  args: width height n_features [q = only the frames that are
not tracked]; PROBE_TURN=k turns the camera around after k frames instead of 24 (the loss comes 41 frames after the
initialisation whatever the content: DESIGN.md section 9)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "send-slam_amd"))
from send_slam_amd import binding, synth
w, h, nf = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (640, 480, 1250)
quiet = len(sys.argv) > 4
sc = synth.scene(4000, w, h)
base = [synth.parallax_frame(4000, w, h, t, sc=sc) for t in range(24)]
order = list(range(24)) + list(range(22, 0, -1))
if os.environ.get("PROBE_TURN"):  # turn around after PROBE_TURN frames instead of 24
    k = int(os.environ["PROBE_TURN"])
    order = list(range(k)) + list(range(k - 2, 0, -1))
cam = binding.Camera(type=b"PinHole", fx=0.8 * w, fy=0.8 * w, cx=w / 2, cy=h / 2, k1=0, k2=0, p1=0, p2=0, width=w, height=h, fps=30, rgb=1,
                     th_depth=40.0, baseline=0.0, depth_map_factor=1000.0)
with binding.OrbContext(0, n_features=nf) as ctx:
    ctx.set_calibration(1, cam)
    for i in range(100):
        r = ctx.track(base[order[i % len(order)]], 1, 1.0 + i / 30.0)
        if quiet and r["state"] == 2:
            continue
        print(i, "content", order[i % len(order)], "state", r["state"], "kp", r["n_keypoints"], "matches", r["n_matches"], "inliers", r["n_inliers"], "map", r["n_map_points"],
              "pos", np.round(r.get("position", [0, 0, 0]), 3) if "position" in r else "")
# with a -DSST_PHASE_TIMING build of the library (SENDSLAM_LIB=...): the pose step's phases over those frames
import ctypes
try:
    lib = binding.load()
    ph = (ctypes.c_double * 8).in_dll(lib, "sst_phase_ms")
    names = ["undistort + arrays", "unique matches", "projection gate", "pose-only", "new points", "history + hand-over"]
    print({n: round(ph[k] / 100, 4) for k, n in enumerate(names)}, "ms per frame")
except ValueError:
    pass
