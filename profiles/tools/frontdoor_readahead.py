"""Front door leg of bench.py at read-ahead 16 / 32 / 64 frames per batch (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
for ra in (16, 32, 64, 16, 32):
    r = bench.bench_frontdoor(readahead=ra)
    print(ra, {k: (v["frames_per_s"], v["frontdoor_seconds"]["submit"], v["frontdoor_seconds"]["recv"], v["frontdoor_seconds"]["track"]) for k, v in r.items() if "frames_per_s" in v}, flush=True)
