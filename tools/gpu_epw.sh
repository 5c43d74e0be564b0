#!/bin/bash
# envs per wavefront of the latency-shaped kernel, re-measured after the ragged-row fix (RISVEC_LAT_EPW = 1 / 2 / 4)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step' % (d['ms_per_step']*1e3))"; }
for rep in 1 2 3; do
 for A in "--config c2 --steps 2000 --warmup 200" "--envs-per-gpu 4096 --ris 40 --steps 2000 --warmup 200" "--envs-per-gpu 2048 --ris 36 --steps 2000 --warmup 200" "--envs-per-gpu 8192 --ris 36 --steps 2000 --warmup 200" "--config c4 --steps 2000 --warmup 200" "--envs-per-gpu 4096 --steps 2000 --warmup 200"; do
  for W in 1 2 4; do
   echo -n "rep $rep [$A] epw=$W: "; RISVEC_LAT_EPW=$W python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
  done
  echo -n "rep $rep [$A] auto: "; python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
 done
done
