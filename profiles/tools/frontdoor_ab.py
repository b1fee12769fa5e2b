import json, os, sys
sys.path.insert(0, "/root/repo")
import bench
for tt in ("0", "1", "0", "1"):
    os.environ["SENDSLAM_TRACK_THREAD"] = tt
    r = bench.bench_frontdoor()
    for k, v in r.items():
        if isinstance(v, dict) and "frames_per_s" in v:
            print("track_thread", tt, k, v["frames_per_s"], v.get("frames_per_s_while_tracking"), v.get("frontdoor_seconds"), flush=True)
