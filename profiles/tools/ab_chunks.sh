#!/bin/bash
# Pipeline throughput (bench.py --timed-only, 300-step rounds) of variant libraries x SENDSLAM_MX_CHUNKS settings: ab_chunks.sh "1 2 4" NAME [NAME ...]
# ("shipped" = the shipped library)
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R"
CH="$1"; shift
for v in "$@"; do
    for c in $CH; do
        if [ "$v" = shipped ]; then unset SENDSLAM_LIB; else export SENDSLAM_LIB="$R/send-slam_amd/lib/libexp_$v.so"; fi
        SENDSLAM_MX_CHUNKS=$c python bench.py --no-cpu-baseline --timed-only --steps 300 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); m=[k for k in j['kernels'] if k['name']=='match'][0]; print('$v chunks $c:', round(j['value']), j['ms_per_step'], 'match live ms', m['mean_ms'])"
    done
done
