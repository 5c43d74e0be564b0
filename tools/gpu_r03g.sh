#!/bin/bash
TAG=${1:-r03g}
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step' % (d['ms_per_step']*1e3))"; }
for M in 20 60 80 100 120; do
  echo -n "32768x8x$M fast: "; python bench.py --envs-per-gpu 32768 --veh 8 --ris $M --no-cpu-baseline --no-legs 2>/dev/null | line
  echo -n "32768x8x$M generic (RISVEC_LAT_EPW=0): "; RISVEC_LAT_EPW=0 python bench.py --envs-per-gpu 32768 --veh 8 --ris $M --no-cpu-baseline --no-legs 2>/dev/null | line
done 2>&1 | tee $OUT/runtime_m_$TAG.txt
cd /tmp && export TMPDIR=/tmp
for RING in 4 6 8; do
  RISVEC_SWEEP_RING=$RING timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5_${TAG}_$RING -o trace -- python3 $R/bench.py --config c5 --steps 300 --warmup 30 --no-cpu-baseline --no-legs > $OUT/prof_c5_${TAG}_$RING.log 2>&1
  echo "ring $RING: $(tail -1 $OUT/prof_c5_${TAG}_$RING.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step' % (d['ms_per_step']*1e3))" 2>/dev/null)"
  grep "sweep8_pair\|k_step_fused_lat" $OUT/prof_c5_${TAG}_$RING/trace_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,150-
done
cd $R
for RING in 4 6 8; do echo -n "c5 ring $RING: "; RISVEC_SWEEP_RING=$RING python bench.py --config c5 --steps 300 --warmup 30 --no-cpu-baseline --no-legs 2>/dev/null | line; done
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +5M -delete
exit 0
