#!/bin/bash
# Small STREAM launches (Netflix-sized periods, a rank's share of an ML-20M period at 8 ranks):
# run length x the library's own choice (--run-len 0).  Output: profiles/r03_sweep_small.txt
cd "$(dirname "$0")/.."
run() { python bench.py --no-cpu-baseline --steps 94 --warmup 10 "$@" 2>&1 | python -c "
import sys, json, os
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-92s step %.4f  kernel %.4f ms  %7.1f M/s' % (' '.join(sys.argv[1:]), j['ms_per_step'], r['kernel_ms_avg'], j['value'] / 1e6))
        break
else:
    print(' '.join(sys.argv[1:]), 'FAILED')
" "$@"; }
for rl in 0 2 4 6 8; do
  run --workload netflix --dim 64 --sampler uniform --run-len $rl
  run --workload netflix --dim 64 --sampler adaptive --refresh-lag 0 --run-len $rl
  run --workload netflix --dim 64 --sampler adaptive --refresh-lag 1 --refresh-cus 64 --run-len $rl
done
for rl in 0 2 4 8; do
  run --workload ml-20m --dim 128 --sampler adaptive --refresh-lag 0 --refresh-split 8 --run-len $rl
  run --workload ml-20m --dim 128 --sampler adaptive --refresh-lag 0 --refresh-split 4 --run-len $rl
done
run --workload ml-20m --dim 128 --sampler adaptive
run --workload ml-20m --dim 128 --sampler adaptive --refresh-lag 0
run --workload ml-20m --dim 128 --sampler uniform
