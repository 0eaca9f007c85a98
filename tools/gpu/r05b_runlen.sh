#!/bin/bash
run() { echo "== $*"; timeout 600 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; st=d.get('steady_state') or {}
print('%8.1f Mtriples/s  step %.4f ms  kernel %.4f ms   steady %.1f M (step %.4f kernel %.4f)' % (d['value']/1e6, d['ms_per_step'], r['kernel_ms_avg'], st.get('value',0)/1e6, st.get('ms_per_step',0), st.get('kernel_ms_avg',0)))"; }
run --run-len 6
run --run-len 12
run --run-len 16
run --hot-rows 512
