#!/bin/bash
run() { echo "== $*"; timeout 900 python bench.py --no-cpu-baseline --workload msd --steady-epochs 0 --sustained-epochs 1 --steps 24 --warmup 4 "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; s=d['config'].get('refresh_schedule',{})
print('%8.1f Mtriples/s  step %.4f ms  kernel %.4f ms  frac %.3f  lag %s cus %s' % (d['value']/1e6, d['ms_per_step'], r['kernel_ms_avg'], r['frac'], s.get('lag'), s.get('side_stream_cus')))"; }
run --refresh-cus 96
run --refresh-cus 64
run --refresh-lag 0
