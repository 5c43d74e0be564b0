#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r02h}
timeout -k 10 900 python -m pytest tests/test_replay_hip.py tests/test_noma_hip.py -m gpu -q -p no:cacheprovider > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $OUT/pytest_$TAG.log | tail -2; grep -E "^(FAILED|ERROR)" $OUT/pytest_$TAG.log | head
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3e env-steps/s  %.2f us/step' % (d['value'], d['ms_per_step']*1e3))"; }
for A in "--replay" "--replay --marshal" "--mode cached --replay" "--mode cached --replay --marshal" "--replay --meter"; do
  echo -n "$A | "; python bench.py $A --steps 1000 --warmup 100 --no-cpu-baseline --no-legs 2>/dev/null | line
done
exit 0
