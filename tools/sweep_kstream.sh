#!/bin/bash
# k_stream levers on one MI355X: heavy-user bitmaps, hot-block size / assignment, CU count.
cd "$(dirname "$0")/.."
run() { python bench.py --no-cpu-baseline --steps 94 --warmup 10 "$@" 2>&1 | python -c "
import sys, json, os
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-34s %-44s step %.4f  k_stream %.4f ms  %7.1f M/s  atomic %.3f' % (tag, ' '.join(sys.argv[1:]), j['ms_per_step'], r['kernel_ms_avg'], j['value'] / 1e6, j.get('roofline_atomic', {}).get('frac', 0)))
        break
else:
    print(tag, ' '.join(sys.argv[1:]), 'FAILED')
" "$@"; }

for t in -1 128 256 512; do BPR_HEAVY_T=$t run; BPR_HEAVY_T=$t run --sampler uniform; done
for h in 256 1024 2048 4096; do run --hot-rows $h --sampler given; BPR_HOT_NAIVE=1 run --hot-rows $h --sampler given; done
for h in 1024 2048; do run --hot-rows $h; done

run; run --hot-rows 1024; run --hot-rows 1024 --sampler given

for c in 224 192 160; do run --main-cus $c --sampler given; run --main-cus $c; done
