#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r02g}
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3e env-steps/s  %.2f us/step  kernel %.2f us' % (d['value'], d['ms_per_step']*1e3, d['roofline']['avg_launch_ms']*1e3))"; }
for A in "--replay" "--mode cached --replay" "--noma" "--replay --meter"; do
  echo -n "$A | "; python bench.py $A --steps 1000 --warmup 100 --no-cpu-baseline --no-legs 2>/dev/null | line
done
cd /tmp && export TMPDIR=/tmp
for NAME in replay cached_replay; do
  [ $NAME = replay ] && A="--replay" || A="--mode cached --replay"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${NAME}_$TAG -o trace -- python3 $R/bench.py $A --steps 500 --warmup 100 --no-cpu-baseline --no-legs > $OUT/prof_${NAME}_$TAG.log 2>&1
  for f in $(find $OUT/prof_${NAME}_$TAG -name "*kernel_stats.csv" | head -1); do head -9 $f | cut -c1-200; done
done
find $OUT -name "*.db" -delete
exit 0
