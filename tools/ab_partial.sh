#!/bin/bash
# the partial snapshot against the full one: step and steady state per CU split of the sorter
run() { python bench.py --steps 48 --warmup 6 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-52s value %.1f M (%.4f ms/step) kernel %.4f | steady %.1f M (%.4f ms/step, kernel %.4f)' % (' '.join(sys.argv[1:]), j['value']/1e6, j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['steady_state']['value']/1e6, j['steady_state']['ms_per_step'], j['steady_state']['kernel_ms_avg']))" "$@"; }
run
run --partial-snapshot 1
run --partial-snapshot 1 --refresh-cus 32
run --partial-snapshot 1 --refresh-cus 32 --partial-target 384
