#!/bin/bash
# round 2, session b: all GPU tests, then small-batch experiments (lat kernel EPW sweep, multi-step launch)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r02b}
timeout -k 10 1100 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $OUT/pytest_$TAG.log | tail -3
grep -E "^(FAILED|ERROR)" $OUT/pytest_$TAG.log | head -20
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3e env-steps/s  %.2f us/step  kernel %.2f us  frac %.3f' % (d['value'], d['ms_per_step']*1e3, d['roofline']['avg_launch_ms']*1e3, d['roofline']['frac']))"; }
for CFG in c2 c4; do
  for EPW in 0 1 2 4; do
    echo -n "$CFG RISVEC_LAT_EPW=$EPW | "; RISVEC_LAT_EPW=$EPW python bench.py --config $CFG --no-cpu-baseline --no-legs 2>/dev/null | line
  done
  for T in 8 32 100; do
    echo -n "$CFG --multi $T | "; python bench.py --config $CFG --multi $T --steps 3200 --no-cpu-baseline --no-legs 2>/dev/null | line
    echo -n "$CFG --multi $T --lean | "; python bench.py --config $CFG --multi $T --steps 3200 --lean --no-cpu-baseline --no-legs 2>/dev/null | line
  done
done
for E in 12288 16384 24576; do
  for EPW in 0 4; do
    echo -n "E=$E RISVEC_LAT_EPW=$EPW | "; RISVEC_LAT_MAX_ENVS=100000 RISVEC_LAT_EPW=$EPW python bench.py --envs-per-gpu $E --no-cpu-baseline --no-legs 2>/dev/null | line
  done
done
echo -n "c3 --multi 16 | "; python bench.py --multi 16 --steps 1600 --no-cpu-baseline --no-legs 2>/dev/null | line
exit 0
