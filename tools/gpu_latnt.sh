#!/bin/bash
# experiment: the latency-shaped kernel (4 envs per wavefront, every request up front) at sizes beyond the Infinity Cache,
# with default and with non-temporal loads (librisvec_ab.so = -DRISVEC_LAT_NT_EXPERIMENT), against the software pipeline
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
ALT=$R/ris_vec_marl_amd/csrc/librisvec_ab.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step  frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))"; }
for rep in 1 2; do
 for E in 65536 131072 262144; do
  A="--envs-per-gpu $E --steps 300 --warmup 30"
  echo -n "rep $rep E=$E pipe (nt auto): "; RISVEC_LAT_MAX_ENVS=0 python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
  echo -n "rep $rep E=$E lat default   : "; RISVEC_LAT_MAX_ENVS=10000000 python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
  echo -n "rep $rep E=$E lat nt        : "; RISVEC_LIB=$ALT RISVEC_LAT_MAX_ENVS=10000000 python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
 done
 echo -n "rep $rep E=32768 pipe: "; RISVEC_LAT_MAX_ENVS=0 python bench.py --no-cpu-baseline --no-legs 2>/dev/null | line
 echo -n "rep $rep E=32768 lat : "; RISVEC_LAT_MAX_ENVS=10000000 python bench.py --no-cpu-baseline --no-legs 2>/dev/null | line
 echo -n "rep $rep E=32768 lat nt: "; RISVEC_LIB=$ALT RISVEC_LAT_MAX_ENVS=10000000 python bench.py --no-cpu-baseline --no-legs 2>/dev/null | line
done
