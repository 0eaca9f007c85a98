#!/bin/bash
# r4: what a rank of an N-rank job runs per step, measured on one GPU (collectives left out)
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_emu; mkdir -p $O
run() { timeout 300 python bench.py --no-cpu-baseline --steps 96 --warmup 8 "$@" 2>$O/err.txt | tail -1 | \
  python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-46s %8.1f Mtriples/s per rank  step %.4f ms  k_stream %.4f ms  chunk %d  sustained %.1f' % (' '.join(sys.argv[1:]), d['value']/1e6, d['ms_per_step'], r['kernel_ms_avg'], d['config']['triples_per_step_per_gpu'], d['sustained']['value']/1e6))" "$@"; }
run
run --emulate-ranks 2
run --emulate-ranks 8
run --emulate-ranks 8 --tier-rows 0
run --emulate-ranks 4 --lr 0.0094
run --emulate-ranks 8 --lr 0.0094
run --emulate-ranks 8 --lr 0.05
run --emulate-ranks 8 --cadence job
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_emu -o bench -- python $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline --emulate-ranks 8 > /tmp/prof_emu.log 2>&1 )
f=$(find /tmp/prof_emu -name "*kernel_trace.csv" | head -1); python tools/timeline.py "$f" > $O/timeline_emu8.txt 2>&1; head -14 $O/timeline_emu8.txt; tail -3 $O/timeline_emu8.txt; tail -3 $O/err.txt
