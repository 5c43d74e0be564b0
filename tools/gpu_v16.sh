#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step' % (d['ms_per_step']*1e3))"; }
for rep in 1 2; do
 for A in "--envs-per-gpu 2048 --veh 16 --ris 256 --mode fused --steps 1000 --warmup 100" "--envs-per-gpu 4096 --veh 16 --ris 256 --mode fused --steps 1000 --warmup 100" "--envs-per-gpu 7168 --veh 16 --ris 256 --mode fused --steps 1000 --warmup 100" "--envs-per-gpu 4096 --veh 16 --ris 64 --steps 2000 --warmup 200" "--envs-per-gpu 8192 --veh 16 --ris 64 --steps 2000 --warmup 200" "--envs-per-gpu 16384 --veh 16 --ris 64 --steps 2000 --warmup 200" "--envs-per-gpu 24576 --veh 16 --ris 64 --steps 2000 --warmup 200"; do
  echo -n "rep $rep [$A] pipe: "; python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
  echo -n "rep $rep [$A] lat : "; RISVEC_LAT_V16=1 python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
 done
done
