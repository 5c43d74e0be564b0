#!/bin/bash
# same-box A/B of the frozen NOMA step: new library vs ris_vec_marl_amd/csrc/librisvec_ab.so (previous build)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
ALT=$R/ris_vec_marl_amd/csrc/librisvec_ab.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step' % (d['ms_per_step']*1e3))"; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_noma_ab.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_noma_ab.log
python tools/profile_noma.py > gpurun_out/noma_new.jsonl 2>&1; RISVEC_LIB=$ALT python tools/profile_noma.py > gpurun_out/noma_old.jsonl 2>&1
echo "--- new"; cat gpurun_out/noma_new.jsonl | cut -c1-300; echo "--- old"; cat gpurun_out/noma_old.jsonl | cut -c1-300
for rep in 1 2 3; do
 for A in "--noma" "--replay" "--mode cached --replay" "--mode cached --noma" "--replay --meter" "--config c5 --noma --steps 200 --warmup 30"; do
  echo -n "rep $rep [$A] new: "; python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
  echo -n "rep $rep [$A] old: "; RISVEC_LIB=$ALT python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
 done
done
