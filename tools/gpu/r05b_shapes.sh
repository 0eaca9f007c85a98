#!/bin/bash
# r5b: the other shapes and options with the binned sort / the lr-aware schedule (bench.py, whole epochs; steady off)
run() { echo "== $*"; timeout 600 python bench.py --no-cpu-baseline --steady-epochs 0 "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; s=d['config'].get('refresh_schedule',{})
print('%8.1f Mtriples/s  step %.4f ms  kernel %.4f ms  frac %.3f  lag %s cus %s' % (d['value']/1e6, d['ms_per_step'], r['kernel_ms_avg'], r['frac'], s.get('lag'), s.get('side_stream_cus')))"; }
run --async-cut 1
run --item-bias 1
run --item-bias 1 --steady-epochs 30
run --workload netflix
run --workload netflix --sampler adaptive
run --workload netflix --sampler adaptive --lr 0.01
run --workload ml-20m --dim 64
run --workload ml-20m --dim 32
run --workload ml-20m --dim 256
run --workload ml-20m --lr 0.05
