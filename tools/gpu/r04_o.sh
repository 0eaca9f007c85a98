#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_o; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_vstream.py -x -q -m gpu ) > $O/pytest_vstream.txt 2>&1; tail -6 $O/pytest_vstream.txt | cut -c1-250
for v in old new; do
  unset BPR_LIB_PATH
  [ $v = old ] && export BPR_LIB_PATH=$R/tools/ubench/variants/libbprcore_old.so
  for opt in adam momentum rmsprop; do
    timeout 600 python bench.py --workload yelp --dim 128 --optimizer $opt --warmup 30 --steps 24 --no-cpu-baseline > $O/$v.$opt.log 2>&1
    tail -1 $O/$v.$opt.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $opt', round(d['value']/1e6,1), 'M triples/s', round(d['ms_per_step'],3), 'ms/step', round(d.get('roofline',{}).get('frac'),4))" 2>&1 | tee -a $O/summary.txt
  done
done
unset BPR_LIB_PATH
for opt in adam; do timeout 600 python bench.py --workload ml-20m --dim 128 --optimizer $opt --steps 48 --no-cpu-baseline > $O/ml20m.$opt.log 2>&1; tail -1 $O/ml20m.$opt.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ml-20m $opt', round(d['value']/1e6,1), 'M triples/s')" | tee -a $O/summary.txt; done
