#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_jit; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "plan_chunk or plan_epoch or pending_split"; BPR_PLAN_CHUNK_SORT=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "plan_chunk" >> $O/pytest.txt 2>&1 > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for j in 1 0; do for r in 1 2; do timeout 300 python bench.py --jit-plan $j --steps 200 --warmup 20 --no-cpu-baseline 2>$O/err_$j.txt | tail -1 > $O/bench_jit${j}_$r.json; done; done
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_jit1_driverlike.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_jit -o bench -- python $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline > /tmp/prof_jit.log 2>&1 )
f=$(find /tmp/prof_jit -name "*kernel_trace.csv" | head -1); python tools/timeline.py "$f" > $O/timeline_jit.txt 2>&1
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_jit/bench_*.json')):
    try:
        j=json.load(open(f)); print(f.split('/')[-1], round(j['value']/1e6,1), 'M/s ms/step', round(j['ms_per_step'],4), 'sustained', round(j['sustained']['value']/1e6,1), j['config']['plan_epoch']['amortised_share_added_ms_per_step'])
    except Exception as ex: print(f, 'ERR', ex)
P
head -12 $O/timeline_jit.txt; tail -3 $O/timeline_jit.txt; tail -3 $O/err_1.txt
timeout 1500 python -m pytest tests/test_gpu_e2e_parity.py -x -q -k "stream_trainer and jit" > $O/pytest_e2e.txt 2>&1; tail -5 $O/pytest_e2e.txt; grep "STREAM\[" $O/pytest_e2e.txt | head -80
# r4 many-seed parity of every STREAM schedule (profiles/e2e_parity_r04.txt)
( python tools/e2e_many_seeds.py sgd adaptive 200 1 strict-own-order,stream-sync,stream-lag1-masked,stream-lag1-masked-jit; python tools/e2e_many_seeds.py sgd uniform 200 1 strict-own-order,stream-sync ) > $O/e2e_many_r04.txt 2>&1; cat $O/e2e_many_r04.txt | cut -c1-220
