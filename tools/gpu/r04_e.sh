#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_e; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_two_tier.py tests/test_gpu_bench.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
run() { timeout 300 python bench.py --no-cpu-baseline --steps 96 --warmup 8 "$@" 2>$O/err.txt | tail -1 | \
  python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-52s %8.1f Mtriples/s per rank  step %.4f ms  k_stream %.4f ms  chunk %d  sustained %.1f' % (' '.join(sys.argv[1:]), d['value']/1e6, d['ms_per_step'], r['kernel_ms_avg'], d['config']['triples_per_step_per_gpu'], d['sustained']['value']/1e6))" "$@"; }
run
run --emulate-ranks 8 --fuse-sync 0
run --emulate-ranks 8
run --emulate-ranks 8
run --emulate-ranks 4 --lr 0.0094
run --emulate-ranks 8 --lr 0.0094
run --emulate-ranks 8 --lr 0.0094 --fuse-sync 0
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_emu -o bench -- python $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline --emulate-ranks 8 > /tmp/prof_emu.log 2>&1 )
f=$(find /tmp/prof_emu -name "*kernel_trace.csv" | head -1); python tools/timeline.py "$f" > $O/timeline_emu8_fused.txt 2>&1; head -12 $O/timeline_emu8_fused.txt; tail -3 $O/timeline_emu8_fused.txt; tail -3 $O/err.txt
timeout 1500 python tools/cadence_study.py --cadence auto --hot-rows 1024 --lr 0.0094 --epochs 20 --eval-every 5 --seeds 10 --ranks 1,4,8 > $O/lr0094_auto_H1024_fused.txt 2>&1; grep "^#" $O/lr0094_auto_H1024_fused.txt | cut -c1-400
