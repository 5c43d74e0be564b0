#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r02l}
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $OUT/pytest_$TAG.log | tail -2; grep -E "^(FAILED|ERROR)" $OUT/pytest_$TAG.log | head
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3e env-steps/s  %.2f us/step  frac %.3f' % (d['value'], d['ms_per_step']*1e3, d['roofline']['frac']))"; }
for A in "" "--config c2" "--config c2 --multi 32 --steps 3200" "--config c4" "--multi 32 --steps 3200" "--mode cached" "--config c5 --steps 300 --warmup 30" "--config big --steps 300 --warmup 30" "--replay" "--mode cached --replay"; do
  echo -n "bench $A | "; python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
done
exit 0
