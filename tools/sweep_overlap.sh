#!/bin/bash
# Overlapped-refresh sweep on one MI355X: bench.py with every snapshot schedule / CU split.
#   tools/sweep_overlap.sh > gpurun_out/sweep_overlap.txt
cd "$(dirname "$0")/.."
run() { python bench.py --no-cpu-baseline --steps 94 --warmup 10 "$@" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']; c = j['config']
        print('%-58s step %.4f ms  k_stream %.4f ms  %7.1f M/s  frac %.3f' % (' '.join(sys.argv[1:]), j['ms_per_step'], r['kernel_ms_avg'], j['value'] / 1e6, r['frac']))
        break
else:
    print(' '.join(sys.argv[1:]), 'FAILED')
" "$@"; }
run --refresh-lag 0
run --refresh-lag 0 --sampler given
run --refresh-lag 0 --sampler uniform
for cus in 0 32 48 64 96 128; do run --refresh-lag 1 --refresh-cus $cus; done
for cus in 0 64 96 128; do run --refresh-lag 1 --refresh-split 2 --refresh-cus $cus; done
for cus in 64 96; do run --refresh-lag 0.4 --refresh-cus $cus; done
run --refresh-lag 1 --refresh-split 3 --refresh-cus 128
