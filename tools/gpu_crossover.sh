#!/bin/bash
# latency-shaped kernel vs software pipeline around the crossover (RISVEC_LAT_MAX_ENVS forces one or the other)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step' % (d['ms_per_step']*1e3))"; }
EL=${1:-"8192 10240 12288 16384 20480"}
for rep in 1 2 3; do
 for M in 36 40 64; do
  for E in $EL; do
   echo -n "rep $rep E=$E M=$M lat: "; RISVEC_LAT_MAX_ENVS=1000000 python bench.py --envs-per-gpu $E --ris $M --steps 2000 --warmup 200 --no-cpu-baseline --no-legs 2>/dev/null | line
   echo -n "rep $rep E=$E M=$M pipe: "; RISVEC_LAT_MAX_ENVS=0 python bench.py --envs-per-gpu $E --ris $M --steps 2000 --warmup 200 --no-cpu-baseline --no-legs 2>/dev/null | line
  done
 done
done
