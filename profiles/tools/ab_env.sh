#!/bin/bash
# bench.py --timed-only under a list of "ENV=V[,ENV=V..][:bench args]" settings, two rounds each, at --steps 20 and --steps 300
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R"
for rep in 1 2; do
for spec in "$@"; do
    envs="${spec%%:*}"; args=""
    if [[ "$spec" == *:* ]]; then args="${spec#*:}"; fi
    for st in 20 300; do
        env $(echo "$envs" | tr ',' ' ') python bench.py --no-cpu-baseline --timed-only --steps $st $args 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$spec', 'steps $st:', round(j['value']), j['ms_per_step'], j['value_spread']['min'], j['value_spread']['max'])"
    done
done
done
