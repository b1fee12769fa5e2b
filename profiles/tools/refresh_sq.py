"""Per-kernel instruction counts of a rocprofv3 --pmc pass, full-batch launches only (all seven resize levels summed).
usage: refresh_sq.py <dir>"""
import collections, csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
by = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "k_" not in n:
        continue
    n = n[n.index("k_"):].split("(")[0].split("<")[0]
    by[n][int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
tot = 0
for n, g in by.items():
    take = sorted(g, reverse=True)[:7 if n == "k_resize_lds" else 1]
    acc = collections.defaultdict(float)
    for x in take:
        for c, v in g[x].items():
            acc[c] += sum(v) / len(v)
    tot += acc.get("SQ_INSTS_VALU", 0)
    print(f"{n:20s}", " ".join(f"{c[3:]}={v / 1e6:.2f}M" for c, v in sorted(acc.items())))
print(f"total VALU {tot / 1e6:.1f} M wave-instructions")
