#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_l; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_vstream.py -x -q -m gpu ) > $O/pytest_vstream.txt 2>&1; tail -15 $O/pytest_vstream.txt | cut -c1-250
python tools/gpu/r04_diag.py 2>&1 | tail -12
for rep in 1; do
for v in old new0 new1; do
  unset BPR_LIB_PATH BPR_VS_DIRECT
  [ $v = old ] && export BPR_LIB_PATH=$R/tools/ubench/variants/libbprcore_old.so
  [ $v = new0 ] && export BPR_VS_DIRECT=0
  for opt in adam momentum sgd; do
    extra=""; [ $opt = sgd ] && extra="--batched"
    timeout 600 python bench.py --workload yelp --dim 128 --optimizer $opt --warmup 30 --steps 24 --no-cpu-baseline $extra > $O/$v.$opt.$rep.log 2>&1
    tail -1 $O/$v.$opt.$rep.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $opt', round(d['value']/1e6,1), 'M triples/s', round(d['ms_per_step'],3), 'ms/step', d.get('roofline',{}).get('frac'))" 2>&1 | tee -a $O/summary.txt
  done
done
done
