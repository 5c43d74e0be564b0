#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))"; }
for A in "--config big --steps 300 --warmup 30" "--config c5 --steps 300 --warmup 30" ""; do
  for ENV in "RISVEC_PIPE_CHUNKED=0" "RISVEC_PIPE_CHUNKED=2"; do
    echo -n "[$A] $ENV (nt auto): "; env $ENV python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
  done
done
for E in 40960 49152 57344 65536; do
  for NT in 0 1; do echo -n "E=$E nt=$NT : "; RISVEC_PIPE_NT=$NT python bench.py --envs-per-gpu $E --steps 500 --warmup 50 --no-cpu-baseline --no-legs 2>/dev/null | line; done
done
for W in 8 12 16; do
  echo -n "big nt waves/CU=$W : "; RISVEC_PIPE_WAVES_PER_CU=$W python bench.py --config big --steps 300 --warmup 30 --no-cpu-baseline --no-legs 2>/dev/null | line
  echo -n "c5 nt waves/CU=$W : "; RISVEC_PIPE_WAVES_PER_CU=$W python bench.py --config c5 --steps 300 --warmup 30 --no-cpu-baseline --no-legs 2>/dev/null | line
done
for D in 1 4; do echo -n "big nt depth=$D : "; RISVEC_PIPE_DEPTH=$D python bench.py --config big --steps 300 --warmup 30 --no-cpu-baseline --no-legs 2>/dev/null | line; done
