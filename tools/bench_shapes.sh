#!/bin/bash
# Throughput of the other BASELINE.json shapes (parity-test cases, not bench lines), with the
# reference's snapshot schedule (--refresh-lag 0) and the overlapped one (lag 1, sort on 64 CUs),
# and the launch sizes an N-rank job runs at the job cadence (period / N triples per launch).
# Usage (GPU box): bash tools/bench_shapes.sh > gpurun_out/shapes.txt
run() { timeout 300 python bench.py --no-cpu-baseline --steps 48 --warmup 6 "$@" 2>&1 | tail -1 | \
  python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-74s %8.1f Mtriples/s  step %.3f ms  kernel %.3f ms  %6.0f GB/s  chunk %d' % (' '.join(sys.argv[1:]), d['value']/1e6, d['ms_per_step'], r['kernel_ms_avg'], r['achieved'], d['config']['triples_per_step_per_gpu']))" "$@"; }
for w in "netflix --dim 64" "ml-20m --dim 128" "msd --dim 256" "yelp --dim 128"; do
  run --workload $w --sampler uniform
  run --workload $w --sampler adaptive --refresh-lag 0
  run --workload $w --sampler adaptive --refresh-lag 1 --refresh-cus 64
  run --workload $w --sampler adaptive   # the default: schedule by shape (fast.auto_schedule)
done
for d in 32 64 256 512; do
  run --workload ml-20m --dim $d --sampler adaptive --refresh-lag 0
  run --workload ml-20m --dim $d --sampler adaptive --refresh-lag 1 --refresh-cus 64
  run --workload ml-20m --dim $d --sampler adaptive
done
echo "# launches of period / N triples (the per-rank step of an N-rank job at the job cadence), refresh between launches"
for n in 2 4 8; do run --workload ml-20m --dim 128 --sampler adaptive --refresh-lag 0 --refresh-split $n; done
echo "# BASELINE configs[4] and the other optimizers through the batched stream"
run --workload yelp --dim 128 --optimizer adam --warmup 30 --steps 24
run --workload yelp --dim 128 --optimizer momentum --steps 24
run --workload yelp --dim 128 --optimizer rmsprop --steps 24
run --workload ml-20m --dim 128 --optimizer adam --steps 48
