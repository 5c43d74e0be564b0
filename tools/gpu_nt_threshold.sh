#!/bin/bash
# where should the latency-shaped kernels switch to non-temporal loads?  RISVEC_LAT_NT=0 (never) vs 1 (always)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step  %.0f MB' % (d['ms_per_step']*1e3, d['roofline']['bytes_per_launch']/1e6))"; }
for rep in 1 2; do
 for E in 6144 8192 10240 12288 16384 20480 24576; do
  A="--envs-per-gpu $E --veh 16 --ris 256 --mode fused --steps 500 --warmup 50"
  echo -n "rep $rep 16x256 E=$E default: "; RISVEC_LAT_NT=0 python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
  echo -n "rep $rep 16x256 E=$E nt     : "; RISVEC_LAT_NT=1 python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
 done
done
