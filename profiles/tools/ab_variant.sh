#!/bin/bash
# A/B of a variant library against the shipped one on the GPU box: ab_variant.sh NAME  (after build_variant.sh NAME ...)
# prints the isolated fast/resize/orient/match times and the bench frames/s of both.
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R"
one() {
    python profiles/tools/time_stages.py 64 30 2>&1 | grep -E "fast|resize|orient|quadtree|match " | tr '\n' ';'
    echo
    python bench.py --no-cpu-baseline --timed-only 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', j['value'], j['ms_per_step'])"
}
export SENDSLAM_LIB="$R/send-slam_amd/lib/libexp_$1.so"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or bit_exact or random_geometries" 2>&1 | tail -1
one "$1"
unset SENDSLAM_LIB
one base
