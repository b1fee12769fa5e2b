#!/bin/bash
# contexts x batch sweep of the default workload on the GPU box (frames/s)
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R"
for i in 1 2; do
  for cb in "3 64" "4 64" "5 64" "6 64" "4 96" "4 128" "3 128"; do
    set -- $cb
    python bench.py --no-cpu-baseline --timed-only --contexts $1 --batch $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ctx', $1, 'batch', $2, round(d['value']))"
  done
done
