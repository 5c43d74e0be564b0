#!/bin/bash
# round 2, session a: parity tests (new + tightened), C2 / C4-shard kernel traces, bench line
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r02a}
timeout -k 10 1000 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $OUT/pytest_$TAG.log | tail -3
grep "parity margin" $OUT/pytest_$TAG.log | sort | awk '{k=$0; sub(/: [0-9.e+-]+ \(worst.*/,"",k); last[k]=$0} END{for(k in last) print last[k]}' | sort > $OUT/margins_$TAG.txt
timeout -k 10 600 python bench.py --steps 2000 --warmup 200 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; tail -2 $OUT/bench_$TAG.err
python bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_gpus2_$TAG.out 2>&1; echo "bench --gpus 2 on one GPU: rc=$? (must be non-zero)"; tail -1 $OUT/bench_gpus2_$TAG.out
cd /tmp && export TMPDIR=/tmp
for CFG in c2 c4; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${CFG}_$TAG -o trace -- python3 $R/bench.py --config $CFG --steps 500 --warmup 50 --no-cpu-baseline --no-legs > $OUT/prof_${CFG}_$TAG.log 2>&1
  echo "rocprof $CFG rc=$?"
  for f in $(find $OUT/prof_${CFG}_$TAG -name "*kernel_stats.csv" | head -1); do head -6 $f | cut -c1-200; done
done
find $OUT -name "*.db" -delete
exit 0
