#!/bin/bash
# C2 / multi probe: repeated runs of the new library only + one kernel trace with per-launch durations
TAG=${1:-probe}
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
ALT=$R/ris_vec_marl_amd/csrc/librisvec_ab.so
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f' % (d['ms_per_step']*1e3), end=' ')"; }
for A in "--config c2" "--config c2 --multi 32 --steps 3200" "--config c4"; do
  echo -n "[$A] new: "; for i in 1 2 3 4 5 6; do python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line; done; echo
  echo -n "[$A] alt: "; for i in 1 2 3; do RISVEC_LIB=$ALT python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line; done; echo
done
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c2_${TAG}_$i -o trace -- python3 $R/bench.py --config c2 --steps 1000 --warmup 100 --no-cpu-baseline --no-legs > $OUT/prof_c2_${TAG}_$i.log 2>&1
grep "k_step_fused_lat" $OUT/prof_c2_${TAG}_$i/trace_kernel_stats.csv | cut -d, -f2-8
done
find $OUT -name "*.db" -delete
exit 0
