#!/bin/bash
# the snapshot schedule at the steady state (bench.py steady_state: one epoch after 30): per workload
run() { python bench.py --steps 24 --warmup 4 --no-cpu-baseline --sustained-epochs 2 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s value %.1f M (%.4f ms/step) kernel %.4f | steady %.1f M (%.4f ms/step, kernel %.4f)' % (' '.join(sys.argv[1:]), j['value']/1e6, j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['steady_state']['value']/1e6, j['steady_state']['ms_per_step'], j['steady_state']['kernel_ms_avg']))" "$@"; }
run --refresh-lag 0
run --refresh-lag 1 --refresh-cus 64
run --refresh-lag 1 --refresh-cus 96
run --refresh-lag 1 --refresh-cus 64 --async-cut 1
run --item-bias 1
run --workload netflix
run --workload netflix --sampler adaptive
run --workload yelp --steady-epochs 10
run --workload yelp --optimizer sgd --steady-epochs 10
