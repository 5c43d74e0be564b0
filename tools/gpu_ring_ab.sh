#!/bin/bash
TAG=${1:-ring}
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests/test_replay_hip.py tests/test_entry_points_hip.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_$TAG.log
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % (d['ms_per_step']*1e3), end=' ')"; }
for A in "--replay" "--mode cached --replay" ""; do
  echo -n "[$A] store in the step kernel: "; for i in 1 2 3; do python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line; done; echo
  echo -n "[$A] separate store launch:    "; for i in 1 2 3; do RISVEC_BENCH_SEPARATE_STORE=1 python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line; done; echo
done 2>&1 | tee $OUT/ring_ab_$TAG.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_replay_$TAG -o trace -- python3 $R/bench.py --replay --steps 500 --warmup 100 --no-cpu-baseline --no-legs > $OUT/prof_replay_$TAG.log 2>&1
grep "risvec" $OUT/prof_replay_$TAG/trace_kernel_stats.csv | cut -d, -f1-4 | cut -c1-90,140- | head -8
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +5M -delete
exit 0
