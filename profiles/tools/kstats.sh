#!/bin/bash
# rocprofv3 kernel-trace statistics of one python tool on the GPU box: kstats.sh TAG script.py [args]  -> prints the per-kernel
# table (name, calls, mean us) and leaves the csv under gpurun_out/TAG_kstats
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
S=$R/$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_kstats -- python3 $S "$@" > $R/gpurun_out/${TAG}_kstats.log 2>&1 || tail -5 $R/gpurun_out/${TAG}_kstats.log
python3 - "$R/gpurun_out/${TAG}_kstats" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s}  mean_us {float(r['AverageNs']) / 1e3:9.2f}  total_ms {float(r['TotalDurationNs']) / 1e6:8.3f}")
PY
