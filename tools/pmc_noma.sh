#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_noma/$tag -o pmc -- python3 $R/tools/profile_noma.py 32768 16 2 > /dev/null 2>$R/gpurun_out/pmc_noma_$tag.err
done
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+'/gpurun_out/pmc_noma/*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name']
        if 'noma' not in k: continue
        agg[k[:60]][row['Counter_Name']].append(float(row['Counter_Value']))
for k,d in agg.items():
    print(k)
    for c,v in sorted(d.items()):
        v=sorted(v); print('   %-24s n=%d median %.4g max %.4g' % (c,len(v),v[len(v)//2],v[-1]))
PY
