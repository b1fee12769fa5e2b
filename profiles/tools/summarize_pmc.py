"""Summarises rocprofv3 --pmc CSVs: mean counter value per launch for every kernel.
usage: summarize_pmc.py <dir with *_counter_collection.csv> [<dir> ...]"""
import collections
import csv
import glob
import json
import sys

out = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            for tag in ("k_", "copy_"):
                if tag in name:
                    name = name[name.index(tag):].split("(")[0].split("<")[0]
                    break
            out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"launches": max(len(v) for v in cs.values())} for k, cs in out.items()}
print(json.dumps(res, indent=1))
