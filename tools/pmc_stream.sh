#!/bin/bash
# SQ counters of k_stream — or of the kernel named by $KERNEL — (run on the GPU box).
# usage: [KERNEL=k_vstream] bash tools/pmc_stream.sh <outdir> [bench args]
out=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_WAIT_ANY" "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /root/repo/gpurun_out/$out/p$i -o x -- python /root/repo/bench.py --no-cpu-baseline --sustained-epochs 0 --steps 12 --warmup 2 "$@" > /dev/null 2>&1
done
python - <<PY
import csv,collections,glob
agg=collections.defaultdict(lambda:[0,0.0])
for f in glob.glob('/root/repo/gpurun_out/$out/p*/**/x_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if '${KERNEL:-k_stream}' in r['Kernel_Name']:
            a=agg[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
for k in sorted(agg): print('%-24s %8d launches  %16.1f per launch' % (k, agg[k][0], agg[k][1]/agg[k][0]))
PY
