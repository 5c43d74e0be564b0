#!/bin/bash
TAG=${1:-sweep}
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_$TAG.log
timeout -k 10 200 python tools/sweep_stamps.py 32768 16 256 > $OUT/sweep_stamps_$TAG.json 2>$OUT/sweep_stamps_$TAG.err; cat $OUT/sweep_stamps_$TAG.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5_$TAG -o trace -- python3 $R/bench.py --config c5 --steps 300 --warmup 30 --no-cpu-baseline --no-legs > $OUT/prof_c5_$TAG.log 2>&1
grep "sweep8_pair\|k_step_fused_lat" $OUT/prof_c5_$TAG/trace_kernel_stats.csv | awk -F'",' '{print substr($1,1,60) " | " $2}'
cd $R
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % (d['ms_per_step']*1e3), end=' ')"; }
echo -n "c5: "; for i in 1 2 3; do python bench.py --config c5 --steps 300 --warmup 30 --no-cpu-baseline --no-legs 2>/dev/null | line; done; echo
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +5M -delete
exit 0
