"""Sums rocprofv3 --pmc counter_collection.csv per kernel (largest grid of each kernel name only).
usage: pmc_summary.py <dir> [name-filter]"""
import collections, csv, glob, os, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "k_"
f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
by = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if flt not in n:
        continue
    n = n[n.index("k_"):].split("(")[0] if "k_" in n else n[:40]
    by[n][int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, g in by.items():
    gs = max(g)
    print(n, "grid", gs, " ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(g[gs].items())))
