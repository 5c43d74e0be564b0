#!/bin/bash
# NOMA grouping (f2): its GPU tests, then the kernel timings at 8 and 16 vehicles (and, when ris_vec_marl_amd/csrc/librisvec_ab.so
# exists, the same timings with that library: same-box A/B).   usage: gpu_noma.sh TAG
TAG=${1:-noma}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
ALT=$R/ris_vec_marl_amd/csrc/librisvec_ab.so
timeout -k 10 900 python -m pytest tests/test_noma_hip.py -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -15 gpurun_out/pytest_$TAG.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/profile_noma.py 32768 8 > gpurun_out/noma_${TAG}_8.jsonl 2>gpurun_out/noma_${TAG}_8.err && cat gpurun_out/noma_${TAG}_8.jsonl | cut -c1-400
for rep in 1 2; do
timeout -k 10 300 python tools/profile_noma.py 32768 16 10 > gpurun_out/noma_${TAG}_16.jsonl 2>gpurun_out/noma_${TAG}_16.err && cat gpurun_out/noma_${TAG}_16.jsonl | cut -c1-400
if [ -f $ALT ]; then echo "--- alt library"; RISVEC_LIB=$ALT timeout -k 10 300 python tools/profile_noma.py 32768 16 10 2>/dev/null | cut -c1-400; fi
done
