#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_acut; mkdir -p $O
run() { timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 "$@" 2>$O/err.txt | tail -1 | \
  python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-46s %8.1f Mtriples/s  step %.4f ms  k_stream %.4f ms  sustained %.1f' % (' '.join(sys.argv[1:]), d['value']/1e6, d['ms_per_step'], r['kernel_ms_avg'], d['sustained']['value']/1e6))" "$@"; }
run --async-cut 1 --refresh-cus 96
run --async-cut 1 --refresh-cus 96
run --async-cut 0 --refresh-cus 96
run --async-cut 1 --refresh-cus 128
run --async-cut 1 --refresh-cus 80
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_acut -o bench -- python $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline --async-cut 1 --refresh-cus 96 > /tmp/prof_acut.log 2>&1 )
f=$(find /tmp/prof_acut -name "*kernel_trace.csv" | head -1); python tools/timeline.py "$f" > $O/timeline_acut96.txt 2>&1; head -5 $O/timeline_acut96.txt; tail -3 $O/timeline_acut96.txt
