#!/bin/bash
run() { echo "== $*"; timeout 600 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; s=d['config'].get('refresh_schedule',{}); st=d.get('steady_state') or {}
print('%8.1f Mtriples/s  step %.4f ms  kernel %.4f ms  frac %.3f  lag %s cus %s  steady %.1f M (step %.4f kernel %.4f)' % (d['value']/1e6, d['ms_per_step'], r['kernel_ms_avg'], r['frac'], s.get('lag'), s.get('side_stream_cus'), st.get('value',0)/1e6, st.get('ms_per_step',0), st.get('kernel_ms_avg',0)))"; }
run --plan-ahead 0
run --plan-ahead 1
run --plan-ahead 1 --async-cut 1
run --plan-ahead 0 --async-cut 1
