#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_m; mkdir -p $O
for v in old v0 v1 old v0 v1; do
  unset BPR_LIB_PATH
  [ $v = old ] && export BPR_LIB_PATH=$R/tools/ubench/variants/libbprcore_old.so
  [ $v = v1 ] && export BPR_LIB_PATH=$R/tools/ubench/variants/libbprcore_v1.so
  for opt in adam momentum; do
    timeout 600 python bench.py --workload yelp --dim 128 --optimizer $opt --warmup 30 --steps 24 --no-cpu-baseline > $O/$v.$opt.log 2>&1
    tail -1 $O/$v.$opt.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $opt', round(d['value']/1e6,1), 'M triples/s', round(d['ms_per_step'],3), 'ms/step')" 2>&1 | tee -a $O/summary.txt
  done
done
