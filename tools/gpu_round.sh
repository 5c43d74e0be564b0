#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel trace (+ optional PMC passes).
# Usage (from the repo root on the GPU box):  bash tools/gpu_round.sh <tag> [pmc]
set -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== pytest -m gpu" | tee $OUT/round_$TAG.log
timeout -k 10 1100 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/pytest_$TAG.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/round_$TAG.log
grep -E "passed|failed" $OUT/pytest_$TAG.log | tail -2 | tee -a $OUT/round_$TAG.log
grep "parity margin" $OUT/pytest_$TAG.log | sed "s/^[.]*//" | sort -u > $OUT/margins_$TAG.txt
echo "== smoke" | tee -a $OUT/round_$TAG.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a $OUT/round_$TAG.log
echo "== bench" | tee -a $OUT/round_$TAG.log
timeout -k 10 600 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
echo "bench rc=$?" | tee -a $OUT/round_$TAG.log
cat $OUT/bench_$TAG.json | tee -a $OUT/round_$TAG.log
tail -3 $OUT/bench_$TAG.err
echo "== rocprofv3 kernel trace" | tee -a $OUT/round_$TAG.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o trace -- python3 $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-legs > $OUT/prof_$TAG.log 2>&1
echo "rocprof rc=$?" | tee -a $OUT/round_$TAG.log
find $OUT/prof_$TAG -name "*stats*.csv" | head -5 | tee -a $OUT/round_$TAG.log
for f in $(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1); do head -12 $f | cut -c1-220 | tee -a $OUT/round_$TAG.log; done
if [ "$2" = "pmc" ]; then
  echo "== rocprofv3 PMC passes (separate runs)" | tee -a $OUT/round_$TAG.log
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${C}_$TAG -o pmc -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-legs > $OUT/pmc_${C}_$TAG.log 2>&1
    echo "pmc $C rc=$?" | tee -a $OUT/round_$TAG.log
  done
  python3 $R/tools/summarize_pmc.py $OUT $TAG 2>&1 | tee -a $OUT/round_$TAG.log
fi
# keep the merged output small
find $OUT/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*.db" -delete
exit 0
