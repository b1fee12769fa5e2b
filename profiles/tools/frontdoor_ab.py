"""Front door leg of bench.py alone, with the tracker thread off and on (GPU box): frames/s, frames/s while tracking, the
front door's stage seconds and the pose step's own split."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
for tt in (sys.argv[1:] or ["0", "1", "0", "1"]):
    os.environ["SENDSLAM_TRACK_THREAD"] = tt
    r = bench.bench_frontdoor()
    for k, v in r.items():
        if isinstance(v, dict) and "frames_per_s" in v:
            print("track_thread", tt, k, v["frames_per_s"], v.get("frames_per_s_while_tracking"), v.get("frontdoor_seconds"), v.get("pose_step_split"), flush=True)
