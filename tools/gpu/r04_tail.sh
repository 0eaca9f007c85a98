#!/bin/bash
# r4: the in-kernel tail cut — parity tests, bench A/B against r3's separate epilogue kernel,
# kernel timeline.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r04_tail; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cut or split or pipeline" > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
for t in 1 0; do
  for r in 1 2; do BPR_TAIL=$t timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_tail${t}_$r.json; done
done
for t in 1 0; do
( cd /tmp && BPR_TAIL=$t timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tail$t -o bench -- python $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline > /tmp/prof_tail$t.log 2>&1 )
f=$(find /tmp/prof_tail$t -name "*kernel_trace.csv" | head -1); python tools/timeline.py "$f" > $O/timeline_tail$t.txt 2>&1
s=$(find /tmp/prof_tail$t -name "*kernel_stats.csv" | head -1); cp "$s" $O/kernel_stats_tail$t.csv
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_tail/bench_*.json')):
    try:
        j=json.load(open(f)); print(f.split('/')[-1], round(j['value']/1e6,1), 'M/s ms/step', round(j['ms_per_step'],4), j['roofline'].get('achieved'))
    except Exception as ex: print(f, 'ERR', ex)
P
tail -3 $O/pytest.txt; for t in 1 0; do echo tail=$t; head -6 $O/timeline_tail$t.txt; tail -3 $O/timeline_tail$t.txt; done
