#!/bin/bash
# One PMC pass (LDS behaviour) over a single-context bench run, on the GPU box: lds_pass.sh TAG -> gpurun_out/TAG_pmc_lds
TAG=${1:-cur}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_lds -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --timed-only --contexts 1 > /dev/null 2>&1
python3 $R/profiles/tools/refresh_sq.py $R/gpurun_out/${TAG}_pmc_lds
