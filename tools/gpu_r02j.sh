#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r02j}
timeout -k 10 600 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; tail -2 $OUT/bench_$TAG.err
python - <<PY
import json
d=json.load(open("$OUT/bench_$TAG.json"))
print("value %.4e  ms/step %.5f  frac %.3f  frac_driver %.3f  yard %.0f GB/s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_driver"], d["roofline"]["yardstick"]["GBps"]))
for k,v in d["legs"].items(): print(k, "%.2f us/step  %.3e env-steps/s  frac %.3f" % (v["ms_per_step"]*1e3, v["env_steps_per_s"], v["roofline_frac"]))
c=d["cpu_baseline"]; print("cpu: all-cores %.0f (%d cores)  1 core %.0f  vectorised %.0f" % (c["value"], c["cores"], c["single_core_value"], c["vectorised_value"]))
PY
RISVEC_DIST_BACKEND=gloo RISVEC_DEVICE_INDEX=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 300 --warmup 30 > $OUT/bench_${TAG}_2rank.json 2> $OUT/bench_${TAG}_2rank.err; echo "2rank rc=$?"; tail -1 $OUT/bench_${TAG}_2rank.json | cut -c1-600; tail -3 $OUT/bench_${TAG}_2rank.err
exit 0
