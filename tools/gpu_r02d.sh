#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r02d}
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "bcd or colsum or edge_shapes or trajectory or determinism or full_size" > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $OUT/pytest_$TAG.log | tail -3
grep -E "^(FAILED|ERROR)" $OUT/pytest_$TAG.log | head -20
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3e env-steps/s  %.2f us/step  kernel %.2f us  frac %.3f' % (d['value'], d['ms_per_step']*1e3, d['roofline']['avg_launch_ms']*1e3, d['roofline']['frac']))"; }
echo -n "c5 generic sweep | "; RISVEC_NO_IDX_SWEEP=1 python bench.py --config c5 --steps 300 --warmup 30 --no-cpu-baseline --no-legs 2>/dev/null | line
echo -n "c5 indexed sweep | "; python bench.py --config c5 --steps 300 --warmup 30 --no-cpu-baseline --no-legs 2>/dev/null | line
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5_$TAG -o trace -- python3 $R/bench.py --config c5 --steps 200 --warmup 20 --no-cpu-baseline --no-legs > $OUT/prof_c5_$TAG.log 2>&1
for f in $(find $OUT/prof_c5_$TAG -name "*kernel_stats.csv" | head -1); do head -6 $f | cut -c1-220; done
find $OUT -name "*.db" -delete; find $OUT/prof_c5_$TAG -name "*kernel_trace.csv" -size +20M -delete
exit 0
