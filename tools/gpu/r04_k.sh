#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_k; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_vstream.py -x -q -m gpu ) > $O/pytest_vstream.txt 2>&1; tail -5 $O/pytest_vstream.txt
( BPR_VS_DIRECT=0 timeout 900 python -m pytest tests/test_gpu_vstream.py -x -q -m gpu ) > $O/pytest_vstream_off.txt 2>&1; tail -3 $O/pytest_vstream_off.txt
for dir in 1 0; do
  for opt in adam momentum rmsprop sgd; do
    extra=""; [ $opt = sgd ] && extra="--batched"
    BPR_VS_DIRECT=$dir timeout 600 python bench.py --workload yelp --dim 128 --optimizer $opt --warmup 30 --steps 24 --no-cpu-baseline $extra > $O/d$dir.$opt.log 2>&1
    tail -1 $O/d$dir.$opt.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('direct=$dir $opt', round(d['value']/1e6,1), 'M triples/s', round(d['ms_per_step'],3), 'ms/step', d.get('roofline',{}).get('frac'))" 2>&1 | tee -a $O/summary.txt
  done
done
BPR_VS_DIRECT=1 timeout 600 python bench.py --workload ml-20m --dim 128 --optimizer adam --steps 48 --no-cpu-baseline > $O/ml20m.adam.log 2>&1; tail -1 $O/ml20m.adam.log | cut -c1-200 | tee -a $O/summary.txt
( time timeout 1500 python -m pytest tests/test_gpu_e2e_parity.py tests/test_gpu_fullscale_parity.py tests/test_gpu_baseline_configs.py -x -q -m gpu -k "batched or cfg5 or adam_batched or vstream" ) > $O/pytest_batched.txt 2>&1; tail -8 $O/pytest_batched.txt
