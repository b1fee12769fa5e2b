#!/bin/bash
# Instruction accounting of k_fast_score: SQ_INSTS_VALU / duration of builds with one phase compiled out (FT_SKIP bits:
# 1 halo ring, 2 blur H, 4 arc search, 8 NMS, 16 blur V, 32 compass + queue, 63 all).  Results of those builds are invalid.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for m in 0 1 2 4 8 16 32 63; do
  L=$R/send-slam_amd/lib/libexp_skip$m.so; [ $m = 0 ] && L=$R/send-slam_amd/lib/libsendslam_orb.so
  export SENDSLAM_LIB=$L
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/acc_$m -- python3 $R/profiles/tools/time_stages.py 64 4 > /dev/null 2>&1
  echo -n "skip $m: "; python3 $R/profiles/tools/pmc_summary.py $R/gpurun_out/acc_$m k_fast | sed 's/k_fast_score grid [0-9]* //'
  python3 $R/profiles/tools/time_stages.py 64 20 2>&1 | grep fast
done
