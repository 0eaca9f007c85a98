cd /tmp && export TMPDIR=/tmp
for s in given uniform adaptive; do
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc_q/$s -o x -- python /root/repo/bench.py --no-cpu-baseline --sustained-epochs 0 --steps 12 --warmup 2 --sampler $s > /dev/null 2>&1
done
python - <<PY
import csv,collections,glob
for s in ("given","uniform","adaptive"):
    agg=collections.defaultdict(lambda:[0,0.0])
    for f in glob.glob('/root/repo/gpurun_out/pmc_q/%s/**/x_counter_collection.csv'%s, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_stream' in r['Kernel_Name']:
                a=agg[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
    print(s, {k: round(v[1]/v[0]/99584,1) for k,v in agg.items()})
PY
