"""Turns the rocprofv3 output directories of the commands in profiles/README.md into the committed summaries.
Only full-batch launches are counted (largest grid of each kernel; for k_resize_lds the seven largest = the seven
levels): the same bench run also launches smaller batches for its latency fields.
usage: refresh_profiles.py <gpurun_out dir> [round tag, default r02] [batch, default 64]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out_dir = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 128
shape = "1280x720_n2000"
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = {"k_ingest_gray16": "ingest", "k_ingest": "ingest", "k_resize_lds": "resize", "k_fast_score": "fast_blur_nms",
         "k_bucket_gather": "bucket_gather", "k_cells_emit": "cells_emit", "k_quadtree": "quadtree", "k_slots": "slots",
         "k_orient_describe": "orient_describe", "k_match": "match", "k_match_mfma": "match", "k_match_mfma_x": "match",
         "k_match_stream": "match_stream"}


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


fetch, write = per_kernel(f"{tag}_pmc_f"), per_kernel(f"{tag}_pmc_w")
raw, traffic = {}, {}
for n in fetch:
    if n not in STAGE or n not in write:
        continue
    f, k = fetch[n][0]["FETCH_SIZE"] / fetch[n][1], write[n][0]["WRITE_SIZE"] / write[n][1]
    raw[n] = {"FETCH_SIZE_KiB_per_launch": round(f, 1), "WRITE_SIZE_KiB_per_launch": round(k, 1), "launches_averaged": fetch[n][1]}
    traffic[n] = int((2 * f + k) * 1024) * fetch[n][1]  # read side x2: profiles/r01_pmc_calibration.json; all levels of the resize summed
sq = {n: dict({c: round(v) for c, v in d.items()}, launches_summed=k) for n, (d, k) in per_kernel(f"{tag}_pmc_sq").items()}
if raw:
    json.dump(raw, open(os.path.join(here, f"{tag}_pmc_fetch_write_per_launch.json"), "w"), indent=1)
if sq:
    json.dump(sq, open(os.path.join(here, f"{tag}_pmc_sq_mix.json"), "w"), indent=1)

# per-FRAME counters read by bench.py (batch-generic): HBM bytes and VALU wave-instructions of full-batch launches / frames
pf_path = os.path.join(here, "per_frame_counters.json")
pf = json.load(open(pf_path)) if os.path.exists(pf_path) else {}
ent = pf.setdefault(shape, {})
for stale in ("k_match_mfma", "k_match_merge"):  # kernels the metric path no longer launches
    ent.pop(stale, None)
for n in set(list(traffic) + list(sq)):
    if n == "k_match_stream":
        continue
    e = {"source": f"{tag} (batch {batch})"}
    if n in traffic:
        e["hbm_bytes_per_frame"] = round(traffic[n] / batch, 1)
    if n in sq and "SQ_INSTS_VALU" in sq[n]:
        e["valu_wave_insts_per_frame"] = round(sq[n]["SQ_INSTS_VALU"] / batch, 1)
    ent[n] = e
json.dump(pf, open(pf_path, "w"), indent=1)

# the database-streaming kernel: own PMC passes over `bench.py --profile-extra match_stream`
sf, sw = per_kernel(f"{tag}_pmc_stream_f"), per_kernel(f"{tag}_pmc_stream_w")
tj_path = os.path.join(here, "traffic.json")
tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
if "k_match_stream" in sf and "k_match_stream" in sw:
    tj["k_match_stream@20M_rows"] = int((2 * sf["k_match_stream"][0]["FETCH_SIZE"] + sw["k_match_stream"][0]["WRITE_SIZE"]) * 1024)
for n, v in traffic.items():
    tj[f"{STAGE[n]}@batch{batch}_{shape}"] = v
json.dump(tj, open(tj_path, "w"), indent=1)

for src, dst in ((f"{tag}_prof4", f"{tag}_bench_kernel_stats.csv"), (f"{tag}_prof1", f"{tag}_bench_kernel_stats_1ctx.csv"),
                 (f"{tag}_prof_stream", f"{tag}_db_stream_kernel_stats.csv"), (f"{tag}_prof_lc", f"{tag}_loop_closure_kernel_stats.csv")):
    f = newest(src, "*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(here, dst))
for src, dst in ((f"{tag}_bench_final.json", f"{tag}_bench.json"), (f"{tag}_bench_steps20.json", f"{tag}_bench_steps20.json"), (f"{tag}_pcie.json", f"{tag}_pcie.json"),
                 (f"{tag}_mfma_probe.txt", f"{tag}_mfma_probe.txt"), (f"{tag}_fp4_probe.txt", f"{tag}_fp4_probe.txt"), (f"{tag}_loop_closure.json", f"{tag}_loop_closure.json"),
                 (f"{tag}_fp4_rate_probe.txt", f"{tag}_fp4_rate_probe.txt"), (f"{tag}_fp4_shape_probe.txt", f"{tag}_fp4_shape_probe.txt")):
    if os.path.exists(os.path.join(out_dir, src)):
        shutil.copy(os.path.join(out_dir, src), os.path.join(here, dst))
tot = sum(v.get("SQ_INSTS_VALU", 0) for v in sq.values())
print(json.dumps({k: v for k, v in tj.items() if f"batch{batch}" in k or "stream" in k}, indent=1))
for n, v in sq.items():
    print(f"{n:20s} VALU {v.get('SQ_INSTS_VALU', 0) / 1e6:7.1f} M  SALU {v.get('SQ_INSTS_SALU', 0) / 1e6:6.1f} M  LDS {v.get('SQ_INSTS_LDS', 0) / 1e6:5.1f} M")
print(f"total VALU wave-instructions per batch: {tot / 1e6:.1f} M -> {tot * 4.05 / (1024 * 2.4e9) * 1e3:.3f} ms at the measured issue rate")
