#!/bin/bash
# same-box A/B of two builds of the library (default vs ris_vec_marl_amd/csrc/librisvec_ab.so); bench arguments in LEGS (one per line)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
ALT=$R/ris_vec_marl_amd/csrc/librisvec_ab.so
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step' % (d['ms_per_step']*1e3))"; }
LEGS=${LEGS:-"--config c2
--config c4"}
for rep in 1 2 3; do
 while IFS= read -r A; do
  echo -n "rep $rep [$A] default: "; python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
  echo -n "rep $rep [$A] alt:     "; RISVEC_LIB=$ALT python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
 done <<< "$LEGS"
done
