#!/bin/bash
# Experiment: streams beyond the Infinity Cache walked in alternating directions (RISVEC_LAT_PINGPONG=1) with the DEFAULT
# cache policy (RISVEC_LAT_NT=0, latency-shaped kernel forced at every size): does an LRU-like cache keep the tail?
TAG=${1:-pp}
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % (d['ms_per_step']*1e3), end=' ')"; }
for E in ${SIZES:-53248 57344 61440 73728 81920 114688}; do
  echo -n "${E}x8x64 as shipped: "; for i in 1 2; do python bench.py --envs-per-gpu $E --no-cpu-baseline --no-legs 2>/dev/null | line; done; echo
  echo -n "${E}x8x64 default policy + pingpong: "; for i in 1 2; do RISVEC_LAT_NT=0 RISVEC_LAT_MAX_ENVS=100000000 RISVEC_LAT_PINGPONG=1 python bench.py --envs-per-gpu $E --no-cpu-baseline --no-legs 2>/dev/null | line; done; echo
done 2>&1 | tee $OUT/pingpong_$TAG.txt
exit 0
