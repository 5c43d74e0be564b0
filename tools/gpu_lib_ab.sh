#!/bin/bash
# Same-box, interleaved A/B of two builds of librisvec.so over the single-launch bench configurations.
#   new = ris_vec_marl_amd/csrc/librisvec.so, alt = ris_vec_marl_amd/csrc/librisvec_ab.so (e.g. built from HEAD)
# Usage: bash tools/gpu_lib_ab.sh <tag> [reps] [skip-tests]
TAG=${1:-ab}; REPS=${2:-3}
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
ALT=$R/ris_vec_marl_amd/csrc/librisvec_ab.so
if [ "$3" != "skip-tests" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_$TAG.log
fi
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f us/step' % (d['ms_per_step']*1e3))"; }
for rep in $(seq 1 $REPS); do
 for A in "--config c2" "--config c4" "--mode cached" "" "--config c2 --multi 32 --steps 3200"; do
  echo -n "rep $rep [$A] new: "; python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
  echo -n "rep $rep [$A] alt: "; RISVEC_LIB=$ALT python bench.py $A --no-cpu-baseline --no-legs 2>/dev/null | line
 done
done 2>&1 | tee $OUT/lib_ab_$TAG.txt
timeout -k 10 200 python tools/lat_stamps.py > $OUT/lat_stamps_$TAG.json 2>$OUT/lat_stamps_$TAG.err; cat $OUT/lat_stamps_$TAG.json
exit 0
