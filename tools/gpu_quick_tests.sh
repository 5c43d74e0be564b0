#!/bin/bash
# full GPU test suite + a few bench lines (one each)
TAG=${1:-qt}
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_$TAG.log
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step  %s' % (d['ms_per_step']*1e3, d['roofline']['kernel']))"; }
for A in "" "--envs-per-gpu 53248" "--envs-per-gpu 65536" "--envs-per-gpu 73728" "--envs-per-gpu 90112" "--config c5 --steps 300 --warmup 30" ${EXTRA:+"$EXTRA"}; do
  echo -n "[$A] "; python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
done
exit 0
