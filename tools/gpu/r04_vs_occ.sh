#!/bin/bash
# k_vstream occupancy sweep (VS_BLOCKS_E4 / VS_LIST_CAP builds in tools/ubench/variants): Yelp shape, d = 128
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_vs_occ; mkdir -p $O
for v in default 3 5 6 8; do
  if [ $v = default ]; then unset BPR_LIB_PATH; else export BPR_LIB_PATH=$R/tools/ubench/variants/libbprcore_vs$v.so; fi
  for opt in adam momentum; do
    timeout 600 python bench.py --workload yelp --dim 128 --optimizer $opt --warmup 30 --steps 24 --no-cpu-baseline > $O/$v.$opt.log 2>&1
    tail -1 $O/$v.$opt.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $opt', round(d['value']/1e6,1), 'M triples/s', round(d['ms_per_step'],3), 'ms/step', d.get('roofline',{}).get('frac'))" 2>&1 | tee -a $O/summary.txt
  done
done
