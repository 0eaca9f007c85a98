#!/bin/bash
# BASELINE configs[3] (MSD shape, d = 256): the snapshot schedule by measurement — serial (lag 0) vs the sort on
# 64 / 96 / 128 masked CUs; the early-state value and the steady state after 10 epochs
run() { python bench.py --workload msd --steps 24 --warmup 4 --no-cpu-baseline --sustained-epochs 2 --steady-epochs 10 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-36s value %.1f M (%.4f ms/step) kernel %.4f ms frac %.3f | steady %.1f M (%.4f ms/step, kernel %.4f)' % (' '.join(sys.argv[1:]), j['value']/1e6, j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['roofline']['frac'], j['steady_state']['value']/1e6, j['steady_state']['ms_per_step'], j['steady_state']['kernel_ms_avg']))" "$@"; }
run --refresh-lag 0
run --refresh-lag 1 --refresh-cus 64
run --refresh-lag 1 --refresh-cus 96
run --refresh-lag 1 --refresh-cus 128
run
run --sampler uniform
