#!/bin/bash
# SQ counters of k_vstream (GPU box), two passes.  usage: bash tools/pmc_vstream.sh [bench args]
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc_vs/p$i -o x -- python /root/repo/bench.py --no-cpu-baseline --sustained-epochs 0 --steps 6 --warmup 30 "$@" > /dev/null 2>&1
done
python - <<PY
import csv,collections,glob
agg=collections.defaultdict(lambda:[0,0.0])
for f in glob.glob('/root/repo/gpurun_out/pmc_vs/p*/**/x_counter_collection.csv', recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if 'k_vstream' in r['Kernel_Name']]
    for r in rows[-6*4:]:
        a=agg[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
for k in sorted(agg): print('%-24s %8d launches  %16.1f per launch' % (k, agg[k][0], agg[k][1]/agg[k][0]))
PY
