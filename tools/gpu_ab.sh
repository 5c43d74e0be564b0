#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
ALT=$R/ris_vec_marl_amd/csrc/librisvec_ab.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step' % (d['ms_per_step']*1e3))"; }
for rep in 1 2 3; do
 for A in "--replay" "--mode cached --replay" "--replay --meter"; do
  echo -n "rep $rep [$A] new: "; python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
  echo -n "rep $rep [$A] alt: "; RISVEC_LIB=$ALT python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
 done
done
