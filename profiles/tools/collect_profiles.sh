#!/bin/bash
# Runs on the GPU box (gpurun): every measurement profiles/ is built from, into gpurun_out/<tag>_*; then
# `python3 profiles/tools/refresh_profiles.py gpurun_out <tag>` (here or in the container) writes the summaries.
# usage: collect_profiles.sh [tag]   (PMC passes are separate runs with --kernel-trace only, as gpurun requires)
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
$B > $O/${TAG}_bench_final.json 2> $O/${TAG}_bench_final.err && echo "bench ok" &&
$B --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_steps20.json 2> $O/${TAG}_bench_steps20.err && echo "bench steps20 ok" &&
export SENDSLAM_BENCH_ROUNDS=1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof4 -- $B --no-cpu-baseline --timed-only --steps 100 > /dev/null 2>&1 && echo "prof4 ok" &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof1 -- $B --no-cpu-baseline --timed-only --contexts 1 --steps 40 > /dev/null 2>&1 && echo "prof1 ok" &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_stream -- $B --profile-extra match_stream > /dev/null 2>&1 && echo "stream ok" &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_lc -- $B --workload loop_closure --steps 5 > $O/${TAG}_loop_closure.json 2> /dev/null && echo "lc ok" &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_f -- $B --steps 3 --warmup 1 --no-cpu-baseline --timed-only --contexts 1 > /dev/null 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_w -- $B --steps 3 --warmup 1 --no-cpu-baseline --timed-only --contexts 1 > /dev/null 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/${TAG}_pmc_sq -- $B --steps 3 --warmup 1 --no-cpu-baseline --timed-only --contexts 1 > /dev/null 2>&1 && echo "pmc ok" &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_stream_f -- $B --profile-extra match_stream > /dev/null 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_stream_w -- $B --profile-extra match_stream > /dev/null 2>&1 && echo "pmc stream ok" &&
python3 $R/profiles/tools/pcie_probe.py > $O/${TAG}_pcie.json 2> /dev/null &&
$R/profiles/tools/mfma_probe > $O/${TAG}_mfma_probe.txt && $R/profiles/tools/fp4_probe > $O/${TAG}_fp4_probe.txt &&
$R/profiles/tools/fp4_rate_probe > $O/${TAG}_fp4_rate_probe.txt && $R/profiles/tools/fp4_shape_probe > $O/${TAG}_fp4_shape_probe.txt && echo "probes ok"
