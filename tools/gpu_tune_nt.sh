#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % (d['ms_per_step']*1e3))"; }
for rep in 1 2 3; do
for W in 6 8 10 12 16; do
  echo -n "rep$rep W=$W big: "; RISVEC_PIPE_WAVES_PER_CU=$W python bench.py --config big --steps 300 --warmup 30 --no-cpu-baseline --no-legs 2>/dev/null | line
  echo -n "rep$rep W=$W c5: "; RISVEC_PIPE_WAVES_PER_CU=$W python bench.py --config c5 --steps 300 --warmup 30 --no-cpu-baseline --no-legs 2>/dev/null | line
done
done
