#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_acut; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "async_cut or cut_at_full or split or pipeline" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for a in 1 0; do for r in 1 2; do timeout 300 python bench.py --async-cut $a --steps 200 --warmup 20 --no-cpu-baseline 2>$O/err_$a.txt | tail -1 > $O/bench_acut${a}_$r.json; done; done
timeout 300 python bench.py --async-cut 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_acut1_driverlike.json
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_acut/bench_*.json')):
    try:
        j=json.load(open(f)); print(f.split('/')[-1], round(j['value']/1e6,1), 'M/s ms/step', round(j['ms_per_step'],4), 'kernel', round(j['roofline']['kernel_ms_avg'],4), 'sustained', round(j['sustained']['value']/1e6,1))
    except Exception as ex: print(f, 'ERR', ex)
P
tail -3 $O/err_1.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_acut -o bench -- python $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline --async-cut 1 > /tmp/prof_acut.log 2>&1 )
f=$(find /tmp/prof_acut -name "*kernel_trace.csv" | head -1); python tools/timeline.py "$f" > $O/timeline_acut.txt 2>&1; head -8 $O/timeline_acut.txt; tail -3 $O/timeline_acut.txt
timeout 1500 python -m pytest tests/test_gpu_e2e_parity.py -x -q -k "stream_trainer and (acut or sync)" > $O/pytest_e2e.txt 2>&1; tail -4 $O/pytest_e2e.txt; grep "acut\] vs\|acut\] adaptive" $O/pytest_e2e.txt | head -20
