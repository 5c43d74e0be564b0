#!/bin/bash
# T-step launch: new library vs ris_vec_marl_amd/csrc/librisvec_ab.so (previous build), same box, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
ALT=$R/ris_vec_marl_amd/csrc/librisvec_ab.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f us/step' % (d['ms_per_step']*1e3))"; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_multi_ab.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_multi_ab.log
for rep in 1 2 3; do
 for A in "--config c2 --multi 32 --steps 3200 --warmup 320" "--config c4 --multi 32 --steps 3200 --warmup 320" "--multi 32 --steps 3200 --warmup 320" "--mode cached --multi 32 --steps 3200 --warmup 320" "--config c5 --mode cached --multi 32 --steps 3200 --warmup 320" "--config c2 --mode cached --multi 32 --steps 3200 --warmup 320"; do
  echo -n "rep $rep [$A] new: "; python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
  echo -n "rep $rep [$A] old: "; RISVEC_LIB=$ALT python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
 done
done
