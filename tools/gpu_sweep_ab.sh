#!/bin/bash
# Same-box A/B of the BCD sweep kernels at BASELINE configs[4]: parity tests, s_memtime stamps of both sweeps,
# then the C5 loop under a kernel trace with the two-lanes-per-env sweep and with the one-lane sweep.
TAG=${1:-r03a}
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_$TAG.log
timeout -k 10 200 python tools/sweep_stamps.py 32768 16 256 pair > $OUT/sweep_stamps_pair_$TAG.json 2>$OUT/sweep_stamps_$TAG.err; cat $OUT/sweep_stamps_pair_$TAG.json
timeout -k 10 200 python tools/sweep_stamps.py 32768 16 256 idx > $OUT/sweep_stamps_idx_$TAG.json 2>>$OUT/sweep_stamps_$TAG.err; cat $OUT/sweep_stamps_idx_$TAG.json
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step' % (d['ms_per_step']*1e3))"; }
for rep in 1 2; do
  echo -n "rep $rep c5 pair: "; python bench.py --config c5 --steps 300 --warmup 30 --no-cpu-baseline --no-legs 2>/dev/null | line
  echo -n "rep $rep c5 one-lane: "; RISVEC_NO_PAIR_SWEEP=1 python bench.py --config c5 --steps 300 --warmup 30 --no-cpu-baseline --no-legs 2>/dev/null | line
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5_$TAG -o trace -- python3 $R/bench.py --config c5 --steps 300 --warmup 30 --no-cpu-baseline --no-legs > $OUT/prof_c5_$TAG.log 2>&1
echo "rocprof c5 rc=$?"
for f in $(find $OUT/prof_c5_$TAG -name "*kernel_stats.csv" | head -1); do head -6 $f | cut -c1-200; done
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +5M -delete
exit 0
