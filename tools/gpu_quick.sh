#!/bin/bash
# quick GPU session: parity tests then bench variants (one line each)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_quick.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_quick.log
run() { echo -n "$* | "; env "$@" python bench.py --no-cpu-baseline $BARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3e env-steps/s  %.1f us/step  %.0f GB/s  frac %.3f' % (d['value'], d['ms_per_step']*1e3, d['roofline']['achieved'], d['roofline']['frac']))"; }
IFS=';' read -ra CFGS <<< "${BENCH_CFGS:-;--envs-per-gpu 8192;--veh 16 --ris 256 --steps 300 --warmup 30}"
for BARGS in "${CFGS[@]}"; do
  echo "== bench $BARGS"
  run RISVEC_NO_PIPE=1
  for w in ${WAVES:-3 4 5 6 8}; do run RISVEC_PIPE_WAVES_PER_CU=$w; done
done
