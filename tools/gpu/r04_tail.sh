#!/bin/bash
# r4, first GPU call: the in-kernel tail cut — parity tests, bench A/B against r3's separate
# epilogue kernel, kernel timeline.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r04_tail; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cut or split or pipeline or stream" > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
for t in 1 0; do
  for r in 1 2 3; do BPR_TAIL=$t timeout 300 python bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 > $O/bench_tail${t}_$r.json; done
done
BPR_TAIL=1 timeout 300 python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/bench_tail1_driverlike.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tail1 -o bench -- python $OLDPWD/bench.py --steps 100 --warmup 10 > /tmp/prof_tail1.log 2>&1 )
f=$(find /tmp/prof_tail1 -name "*kernel_trace.csv" | head -1); python tools/timeline.py $f > $O/timeline_tail1.txt 2>&1
s=$(find /tmp/prof_tail1 -name "*kernel_stats.csv" | head -1); cp $s $O/kernel_stats_tail1.csv
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_tail/bench_*.json')):
    try:
        j=json.load(open(f)); print(f.split('/')[-1], round(j['value']/1e6,1), 'M/s ms/step', j['ms_per_step'], 'kernel', j['roofline'].get('kernel_ms'))
    except Exception as ex: print(f, 'ERR', ex)
P
tail -5 $O/pytest.txt; cat $O/timeline_tail1.txt | tail -4
