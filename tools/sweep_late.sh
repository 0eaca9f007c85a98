#!/bin/bash
# k_stream variants on one MI355X: late atomics x look-ahead x occupancy build.
cd "$(dirname "$0")/.."
run() { python bench.py --no-cpu-baseline --steps 94 --warmup 10 "$@" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-40s step %.4f ms  k_stream %.4f ms  %7.1f M/s  atomic frac %.3f' % (' '.join(sys.argv[1:]), j['ms_per_step'], r['kernel_ms_avg'], j['value'] / 1e6, j.get('roofline_atomic', {}).get('frac', 0)))
        break
else:
    print(' '.join(sys.argv[1:]), 'FAILED')
" "$@"; }
for lib in "" _w4; do for late in 0 1; do for look in 0 6; do
  echo "lib=libbprcore$lib.so late=$late look=$look"
  [ -n "$lib" ] && export BPR_LIB_PATH=$PWD/revisit-bpr_amd/libbprcore$lib.so || unset BPR_LIB_PATH
  export BPR_STREAM_LATE=$late BPR_STREAM_LOOK=$look
  run
  run --sampler given
  [ "$look" = 6 ] && run --refresh-lag 1 --refresh-cus 64
done; done; done
