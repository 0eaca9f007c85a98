#!/bin/bash
# Memory-side counters of k_stream (GPU box): where do the atomics execute, what stalls the L2?
# usage: bash tools/pmc_mem.sh <outdir> [bench args]      (counters collected in separate --pmc passes)
out=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_REQ TCC_READ_REQ TCC_WRITE_REQ TCC_ATOMIC" \
           "TCC_EA0_ATOMIC TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_EA0_WRREQ_ATOMIC_DRAM" \
           "TCC_RW_ATOMIC_REQ TCC_NC_ATOMIC_REQ TCC_UC_ATOMIC_REQ TCC_CC_ATOMIC_REQ" \
           "TCC_TAG_STALL TCC_BUSY TCC_HIT TCC_MISS" \
           "TCC_EA0_ATOMIC_LEVEL TCC_EA0_RDREQ_LEVEL TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ_STALL" \
           "TCP_TCC_READ_REQ TCP_TCC_ATOMIC_WITHOUT_RET_REQ TCP_PENDING_STALL_CYCLES TCP_TOTAL_ACCESSES" \
           "TCP_TCC_READ_REQ_LATENCY TCP_TCC_WRITE_REQ_LATENCY TCP_ATOMIC_TAGCONFLICT_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES" \
           "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_ATOMIC_WAVEFRONTS" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /root/repo/gpurun_out/$out/p$i -o x -- python /root/repo/bench.py --no-cpu-baseline --sustained-epochs 0 --steps 12 --warmup 2 "$@" > /root/repo/gpurun_out/$out.p$i.log 2>&1
done
python - <<PY
import csv,collections,glob
agg=collections.defaultdict(lambda:[0,0.0])
for f in glob.glob('/root/repo/gpurun_out/$out/p*/**/x_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_stream<' in r['Kernel_Name']:
            a=agg[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
for k in sorted(agg): print('%-40s %6d launches  %18.1f per launch' % (k, agg[k][0], agg[k][1]/agg[k][0]))
PY
