#!/bin/bash
# Full end-of-round GPU session: gpu_round.sh (tests, smoke, bench, kernel trace, PMC passes) plus the
# per-kernel table, the NOMA timings and the other BASELINE configs.  Usage: bash tools/gpu_full.sh <tag>
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd $R
bash tools/gpu_round.sh $TAG pmc || exit 1
echo "== every kernel at the C3 and C5 shapes" | tee -a $OUT/round_$TAG.log
cd /tmp && export TMPDIR=/tmp
for SHAPE in "32768 8 64" "32768 16 256"; do
  N=$(echo $SHAPE | tr ' ' '_')
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/allk_${TAG}_$N -o trace -- python3 $R/tools/profile_all_kernels.py $SHAPE 10 > $OUT/allk_${TAG}_$N.json 2> $OUT/allk_${TAG}_$N.err
  echo "allk $SHAPE rc=$?" | tee -a $OUT/round_$TAG.log
  tail -1 $OUT/allk_${TAG}_$N.json > $OUT/allk_${TAG}_${N}_meta.json
  python3 $R/tools/kernel_table.py $(find $OUT/allk_${TAG}_$N -name "*kernel_stats.csv" | head -1) $OUT/allk_${TAG}_${N}_meta.json | tee $OUT/allk_${TAG}_$N.md
done
cd $R
echo "== NOMA grouping timings" | tee -a $OUT/round_$TAG.log
timeout -k 10 300 python tools/profile_noma.py 32768 8 > $OUT/noma_${TAG}_8.json 2>/dev/null; cat $OUT/noma_${TAG}_8.json
timeout -k 10 300 python tools/profile_noma.py 32768 16 10 > $OUT/noma_${TAG}_16.json 2>/dev/null; cat $OUT/noma_${TAG}_16.json
timeout -k 10 200 python tools/noma_stamps.py > $OUT/noma_stamps_$TAG.jsonl 2>/dev/null; cut -c1-300 $OUT/noma_stamps_$TAG.jsonl
echo "== other configs" | tee -a $OUT/round_$TAG.log
b() { NAME=$1; shift; timeout -k 10 400 python bench.py --no-cpu-baseline --no-legs "$@" > $OUT/bench_${TAG}_$NAME.json 2>/dev/null; python3 -c "import json,sys; d=json.load(open('$OUT/bench_${TAG}_$NAME.json')); print('$NAME: %.3e env-steps/s  %.1f us/step  frac %.3f' % (d['value'], d['ms_per_step']*1e3, d['roofline']['frac']))"; }
b c2 --config c2
b c2_multi32 --config c2 --multi 32 --steps 3200
b c4shard --config c4
b c4shard_multi32 --config c4 --multi 32 --steps 3200
b c3_multi32 --multi 32 --steps 3200
b big --config big --steps 300 --warmup 30
b c5 --config c5 --steps 300 --warmup 30
b cached --mode cached
b cached_multi32 --mode cached --multi 32 --steps 3200
b c5_cached_multi32 --config c5 --mode cached --multi 32 --steps 3200
b noma --noma
b cached_noma --mode cached --noma
b replay --replay
b replay_marshal --replay --marshal
b cached_replay --mode cached --replay
b sarl --mode sarl
b policy --policy --steps 300 --warmup 30
b steer --steer
b c5_steer --config c5 --steps 300 --warmup 30 --steer
b replay_steer --replay --steer
b meter --meter
b replay_meter --replay --meter
echo "== kernel traces at the other BASELINE configs" | tee -a $OUT/round_$TAG.log
cd /tmp && export TMPDIR=/tmp
for CFG in c2 c4 c5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${CFG}_$TAG -o trace -- python3 $R/bench.py --config $CFG --steps 300 --warmup 30 --no-cpu-baseline --no-legs > $OUT/prof_${CFG}_$TAG.log 2>&1
  echo "rocprof $CFG rc=$?" | tee -a $OUT/round_$TAG.log
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c2multi_$TAG -o trace -- python3 $R/bench.py --config c2 --multi 32 --steps 3200 --no-cpu-baseline --no-legs > $OUT/prof_c2multi_$TAG.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_replay_$TAG -o trace -- python3 $R/bench.py --replay --steps 500 --warmup 100 --no-cpu-baseline > $OUT/prof_replay_$TAG.log 2>&1
cd $R
python tools/sweep_stamps.py 32768 16 256 > $OUT/sweep_stamps_$TAG.json 2>/dev/null; cat $OUT/sweep_stamps_$TAG.json
timeout -k 10 200 python tools/lat_stamps.py > $OUT/lat_stamps_$TAG.json 2>/dev/null; cat $OUT/lat_stamps_$TAG.json
echo "== streaming yardstick" | tee -a $OUT/round_$TAG.log
timeout -k 10 300 python tools/membench.py > $OUT/membench_$TAG.jsonl 2>/dev/null; cat $OUT/membench_$TAG.jsonl
echo "== 2-rank rehearsal on one GPU (gloo for the collective; RCCL needs one GPU per rank)" | tee -a $OUT/round_$TAG.log
RISVEC_DIST_BACKEND=gloo RISVEC_DEVICE_INDEX=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 300 --warmup 30 --no-cpu-baseline \
  > $OUT/bench_${TAG}_2rank.json 2> $OUT/bench_${TAG}_2rank.err; echo "2rank rc=$?"; tail -1 $OUT/bench_${TAG}_2rank.json | cut -c1-400
find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -size +5M -delete
exit 0
