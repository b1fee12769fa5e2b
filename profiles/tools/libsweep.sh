# usage: libsweep.sh <lib-suffix>...   -- match stage time (one context) and frames/s of bench.py for alternative builds
for i in 1 2; do for v in "$@"; do
  L=$PWD/send-slam_amd/lib/libsendslam_orb$v.so
  SENDSLAM_LIB=$L python profiles/tools/time_match.py 2>/dev/null | tail -1
  SENDSLAM_LIB=$L python profiles/tools/bench_with_lib.py --no-cpu-baseline --timed-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib$v', d['value'])"
done; done
