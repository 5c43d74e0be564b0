#!/bin/bash
# same-box A/B of the ragged-row load fix (M = 36 / 40): new library vs ris_vec_marl_amd/csrc/librisvec_ab.so
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
ALT=$R/ris_vec_marl_amd/csrc/librisvec_ab.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step  frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))"; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_ragged.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_ragged.log
for rep in 1 2 3; do
 for A in "--config c2 --steps 2000 --warmup 200" "--config c2 --multi 32 --steps 3200 --warmup 320" "--envs-per-gpu 32768 --ris 40" "" "--config c4 --steps 2000 --warmup 200" "--config c4 --multi 32 --steps 3200 --warmup 320" "--mode cached --steps 2000 --warmup 200" "--mode cached --replay" "--envs-per-gpu 4096 --ris 40 --steps 2000 --warmup 200" "--envs-per-gpu 2048 --steps 2000 --warmup 200" "--envs-per-gpu 1024 --veh 4 --ris 16 --steps 2000 --warmup 200"; do
  echo -n "rep $rep [$A] new: "; python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
  echo -n "rep $rep [$A] old: "; RISVEC_LIB=$ALT python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
 done
done
