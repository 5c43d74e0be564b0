#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f us/step  %s' % (d['ms_per_step']*1e3, d['roofline']['kernel']))"; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_cmulti.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_cmulti.log
for A in "--mode cached --steps 2000 --warmup 200" "--mode cached --multi 32 --steps 3200 --warmup 320" "--config c2 --mode cached --steps 2000 --warmup 200" "--config c2 --mode cached --multi 32 --steps 3200 --warmup 320" "--config c5 --mode cached --steps 1000 --warmup 100" "--config c5 --mode cached --multi 32 --steps 3200 --warmup 320" "--config c5 --mode fused --multi 32 --steps 320 --warmup 32" "--envs-per-gpu 32768 --veh 16 --ris 64 --multi 32 --steps 3200 --warmup 320" "--config c2 --multi 32 --steps 3200 --warmup 320"; do
  echo -n "[$A]: "; python bench.py $A --no-cpu-baseline --no-legs 2>&1 | tail -1 | line
done
