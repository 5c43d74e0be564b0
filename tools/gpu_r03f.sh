#!/bin/bash
TAG=${1:-r03f}
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_$TAG.log
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step' % (d['ms_per_step']*1e3))"; }
for rep in 1 2; do
  echo -n "rep $rep c5 lazy theta: "; python bench.py --config c5 --steps 300 --warmup 30 --no-cpu-baseline --no-legs 2>/dev/null | line
  echo -n "rep $rep c5 eager theta: "; RISVEC_BENCH_EAGER_THETA=1 python bench.py --config c5 --steps 300 --warmup 30 --no-cpu-baseline --no-legs 2>/dev/null | line
done
for M in 20 60 80 120; do
  echo -n "32768x8x$M fast: "; python bench.py --envs-per-gpu 32768 --veh 8 --ris $M --no-cpu-baseline --no-legs 2>/dev/null | line
  echo -n "32768x8x$M generic (HEAD~ lib): "; RISVEC_LIB=$R/ris_vec_marl_amd/csrc/librisvec_ab.so python bench.py --envs-per-gpu 32768 --veh 8 --ris $M --no-cpu-baseline --no-legs 2>/dev/null | line
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5_$TAG -o trace -- python3 $R/bench.py --config c5 --steps 300 --warmup 30 --no-cpu-baseline --no-legs > $OUT/prof_c5_$TAG.log 2>&1
echo "rocprof c5 rc=$?"
for f in $(find $OUT/prof_c5_$TAG -name "*kernel_stats.csv" | head -1); do grep "risvec" $f | cut -c1-220 | head -5; done
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +5M -delete
exit 0
