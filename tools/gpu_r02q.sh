#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_r02q.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $OUT/pytest_r02q.log | tail -2; grep -E "^(FAILED|ERROR)" $OUT/pytest_r02q.log | head
python tools/colsum_time.py; RISVEC_COLSUM_NT=0 python tools/colsum_time.py
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))"; }
for A in "" "--config big --steps 300 --warmup 30" "--config c5 --steps 300 --warmup 30" "--mode sarl --envs-per-gpu 262144 --steps 200 --warmup 20"; do echo -n "[$A] : "; python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line; done
