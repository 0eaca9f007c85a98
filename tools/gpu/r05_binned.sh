#!/bin/bash
# r5: the binned snapshot sort in the timed schedule — the CU split re-measured (bench.py, ML-20M shape)
out=gpurun_out/r05_binned_bench.txt; : > $out
for cfg in "--refresh-cus 64" "--refresh-cus 32" "--refresh-lag 0" ; do
  echo "== $cfg" >> $out
  python bench.py --no-cpu-baseline $cfg 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    j=json.loads(line)
    print({k:j.get(k) for k in ('value','ms_per_step')}, 'twenty_steps', j.get('timed_region',{}).get('value') if isinstance(j.get('timed_region'),dict) else None, 'steady', (j.get('steady_state') or {}).get('value'), 'roofline', j['roofline'].get('frac'), 'kernel_ms', j['roofline'].get('kernel_ms'))
" >> $out
done
cat $out
