#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_v; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_config_run.py tests/test_gpu_two_tier.py tests/test_gpu_example.py -x -q -m gpu ) > $O/pytest_bias.txt 2>&1; tail -5 $O/pytest_bias.txt
for b in 1 0 1 0; do
  timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --sustained-epochs 0 --item-bias $b > $O/b$b.log 2>&1
  tail -1 $O/b$b.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('item_bias=$b', round(d['value']/1e6,1), 'M triples/s', round(d['ms_per_step'],4), 'ms/step', round(d['roofline']['frac'],4))" 2>&1 | tee -a $O/summary.txt
done
