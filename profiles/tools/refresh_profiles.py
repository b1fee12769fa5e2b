"""Turns the rocprofv3 output directories of the commands in profiles/README.md into the committed summaries.
Only full-batch launches are counted (largest grid of each kernel; for k_resize_lds the seven largest = the seven
levels): the same bench run also launches single-frame batches for its latency fields.
usage: refresh_profiles.py <gpurun_out dir> [batch-key]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out_dir = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "batch64_1280x720_n2000"
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = {"k_ingest_gray16": "ingest", "k_ingest": "ingest", "k_resize_lds": "resize", "k_fast_score": "fast_blur_nms",
         "k_bucket_gather": "bucket_gather", "k_cells_emit": "cells_emit", "k_quadtree": "quadtree", "k_slots": "slots",
         "k_orient_describe": "orient_describe", "k_match": "match", "k_match_mfma": "match"}


def newest(d, pattern):
    """gpurun merges new files into gpurun_out/ without removing old ones: take the latest run only"""
    f = glob.glob(os.path.join(out_dir, d, "**", pattern), recursive=True)
    return max(f, key=os.path.getmtime) if f else None


def rows(d):
    f = newest(d, "*counter_collection.csv")
    return list(csv.DictReader(open(f))) if f else []


def per_kernel(d):
    by = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
    for r in rows(d):
        n = r["Kernel_Name"]
        if "k_" not in n:
            continue
        n = n[n.index("k_"):].split("(")[0].split("<")[0]
        by[n][int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for n, g in by.items():
        take = sorted(g, reverse=True)[:7 if n == "k_resize_lds" else 1]
        acc = collections.defaultdict(float)
        for x in take:
            for c, v in g[x].items():
                acc[c] += sum(v) / len(v)
        res[n] = (dict(acc), len(take))
    return res


fetch, write = per_kernel("pmc_f"), per_kernel("pmc_w")
raw, traffic = {}, {}
for n in fetch:
    if n not in STAGE or n not in write:
        continue
    f, k = fetch[n][0]["FETCH_SIZE"] / fetch[n][1], write[n][0]["WRITE_SIZE"] / write[n][1]
    raw[n] = {"FETCH_SIZE_KiB_per_launch": round(f, 1), "WRITE_SIZE_KiB_per_launch": round(k, 1)}
    traffic[f"{STAGE[n]}@{key}"] = int((2 * f + k) * 1024)  # read side x2: profiles/r01_pmc_calibration.json
json.dump(traffic, open(os.path.join(here, "traffic.json"), "w"), indent=1)
json.dump(raw, open(os.path.join(here, "r01_pmc_fetch_write_per_launch.json"), "w"), indent=1)
sq = {n: dict({c: round(v) for c, v in d.items()}, launches_summed=k) for n, (d, k) in per_kernel("pmc_sq").items()}
json.dump(sq, open(os.path.join(here, "r01_pmc_sq_mix.json"), "w"), indent=1)
for src, dst in (("prof4", "r01_bench_kernel_stats.csv"), ("prof1", "r01_bench_kernel_stats_1ctx.csv")):
    f = newest(src, "*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(here, dst))
if os.path.exists(os.path.join(out_dir, "bench_final.json")):
    shutil.copy(os.path.join(out_dir, "bench_final.json"), os.path.join(here, "r01_bench.json"))
tot = sum(v.get("SQ_INSTS_VALU", 0) for v in sq.values())
print(json.dumps(traffic, indent=1))
for n, v in sq.items():
    print(f"{n:20s} VALU {v.get('SQ_INSTS_VALU', 0) / 1e6:7.1f} M  SALU {v.get('SQ_INSTS_SALU', 0) / 1e6:6.1f} M  LDS {v.get('SQ_INSTS_LDS', 0) / 1e6:5.1f} M")
print(f"total VALU wave-instructions per batch: {tot / 1e6:.1f} M -> {tot * 4.05 / (1024 * 2.4e9) * 1e3:.3f} ms at the measured issue rate")
