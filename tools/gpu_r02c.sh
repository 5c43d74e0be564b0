#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r02c}
timeout -k 10 1100 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $OUT/pytest_$TAG.log | tail -3
grep -E "^(FAILED|ERROR)" $OUT/pytest_$TAG.log | head -20
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3e env-steps/s  %.2f us/step  kernel %.2f us  frac %.3f' % (d['value'], d['ms_per_step']*1e3, d['roofline']['avg_launch_ms']*1e3, d['roofline']['frac']))"; }
for CFG in c2 c4 c3; do
  for EPW in 1 2 4 8; do
    echo -n "$CFG --multi 32 RISVEC_MULTI_EPW=$EPW | "; RISVEC_MULTI_EPW=$EPW python bench.py --config $CFG --multi 32 --steps 3200 --no-cpu-baseline --no-legs 2>/dev/null | line
  done
done
echo -n "c3 single | "; python bench.py --no-cpu-baseline --no-legs 2>/dev/null | line
exit 0
