#!/bin/bash
# GPU_MAX_HW_QUEUES x contexts sweep of the default workload on the GPU box (frames/s)
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R"
for q in 4 8; do
  for c in 4 5 6 8; do
    GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --timed-only --contexts $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hwq', $q, 'ctx', $c, round(d['value']))"
  done
done
