#!/bin/bash
# same-box A/B against the round-2 tree (ab_r02/: package + bench.py of commit 0af1425, built in the container)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % (d['ms_per_step']*1e3), end=' ')"; }
for A in "--config big --steps 300 --warmup 30" "--config c5 --steps 300 --warmup 30" "--envs-per-gpu 65536" "--envs-per-gpu 131072 --steps 500 --warmup 50" ${EXTRA:+"$EXTRA"}; do
  echo -n "[$A] new: "; for i in 1 2 3; do python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line; done; echo
  echo -n "[$A] r02: "; for i in 1 2 3; do (cd ab_r02 && python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null) | line; done; echo
done 2>&1 | tee $OUT/r02_ab_${1:-x}.txt
exit 0
