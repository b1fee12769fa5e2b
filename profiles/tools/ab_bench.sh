#!/bin/bash
# Alternating bench runs of variant libraries against the shipped one on the GPU box: ab_bench.sh NAME [NAME ...]
# (after build_variant.sh NAME ...); three rounds, frames/s and ms/step per run.
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R"
run() {
    python bench.py --no-cpu-baseline --timed-only 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', round(j['value']), j['ms_per_step'])"
}
for rep in 1 2 3; do
    for v in "$@"; do
        SENDSLAM_LIB="$R/send-slam_amd/lib/libexp_$v.so" run "$v"
    done
    run shipped
done
