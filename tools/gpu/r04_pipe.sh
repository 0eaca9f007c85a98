#!/bin/bash
# r4: the pipelined adaptive sampler — oracle parity, bench, kernel time
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_pipe; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_api.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
for r in 1 2 3; do timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$r.json; done
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --refresh-lag 0 2>/dev/null | tail -1 > $O/bench_lag0.json
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_pipe/bench_*.json')):
    try:
        j=json.load(open(f)); print(f.split('/')[-1], round(j['value']/1e6,1), 'M/s ms/step', round(j['ms_per_step'],4), 'kernel', round(j['roofline']['kernel_ms_avg'],4), 'sustained', round(j['sustained']['value']/1e6,1))
    except Exception as ex: print(f, 'ERR', ex)
P
