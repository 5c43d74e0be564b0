#!/bin/bash
# Copy what a tools/gpu_full.sh session left under gpurun_out/ into profiles/ (tracked) under the names DESIGN.md cites.
# usage: bash tools/collect_profiles.sh TAG
T=$1; O=gpurun_out; P=profiles
cp $O/bench_$T.json $P/${T}_bench.json
for f in $O/bench_${T}_*.json; do n=$(basename $f .json); n=${n#bench_${T}_}; [ "$n" = 2rank ] && n=2rank_one_gpu; cp $f $P/${T}_bench_$n.json; done
cp $O/prof_$T/trace_kernel_stats.csv $P/${T}_kernel_stats.csv
for c in c2 c4 c5 c2multi replay; do [ -f $O/prof_${c}_$T/trace_kernel_stats.csv ] && cp $O/prof_${c}_$T/trace_kernel_stats.csv $P/${T}_${c}_kernel_stats.csv; done
for n in 32768_8_64 32768_16_256; do
  cp $O/allk_${T}_$n.md $P/${T}_all_kernels_$n.md
  cp $(find $O/allk_${T}_$n -name "*kernel_stats.csv" | head -1) $P/${T}_all_kernels_${n}_stats.csv
done
cp $O/pmc_summary_$T.json $P/${T}_pmc_summary.json
cp $O/margins_$T.txt $P/${T}_parity_margins.txt
cat $O/noma_${T}_8.json $O/noma_${T}_16.json > $P/${T}_noma_timings.jsonl
cp $O/noma_stamps_$T.jsonl $P/${T}_noma_stamps.jsonl
cp $O/sweep_stamps_$T.json $P/${T}_sweep_stamps.json
cp $O/lat_stamps_$T.json $P/${T}_c2_lat_stamps.json
cp $O/membench_$T.jsonl $P/${T}_membench.jsonl
ls $P | grep -c "^${T}_"
