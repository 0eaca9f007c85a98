#!/bin/bash
# is k_stream sensitive to its code size?  512 executed s_nop per step as 2 KB of straight code vs a 32-byte loop
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_p; mkdir -p $O
for v in base padloop padcode base padloop padcode; do
  unset BPR_LIB_PATH
  [ $v != base ] && export BPR_LIB_PATH=$R/tools/ubench/variants/libbprcore_$v.so
  timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --sustained-epochs 0 > $O/$v.log 2>&1
  tail -1 $O/$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,1), 'M triples/s', round(d['ms_per_step'],4), 'ms/step', round(d['roofline']['frac'],4))" 2>&1 | tee -a $O/summary.txt
done
