#!/bin/bash
# One PMC pass (instruction mix) over a single-context bench run, on the GPU box: sq_pass.sh TAG -> gpurun_out/TAG_pmc_sq
TAG=${1:-cur}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --timed-only --contexts 1 > /dev/null 2>&1
python3 $R/profiles/tools/refresh_sq.py $R/gpurun_out/${TAG}_pmc_sq
