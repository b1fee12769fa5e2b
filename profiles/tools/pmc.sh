#!/bin/bash
# One PMC pass of a python tool on the GPU box: pmc.sh TAG "COUNTER COUNTER ..." script.py [args] -> per-kernel means of every counter
# (kernel-trace only beside --pmc, as gpurun requires); csv under gpurun_out/TAG_pmc
TAG=$1; CTRS=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
S=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc -- python3 $S "$@" > $R/gpurun_out/${TAG}_pmc.log 2>&1 || tail -5 $R/gpurun_out/${TAG}_pmc.log
python3 - "$R/gpurun_out/${TAG}_pmc" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:32s} n {len(v):4d}  mean {sum(v) / len(v):16.1f}")
PY
