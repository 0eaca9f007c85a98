#!/bin/bash
# Throughput of the other BASELINE.json shapes (parity-test cases, not bench lines).
# Usage (GPU box): bash tools/bench_shapes.sh > gpurun_out/shapes.txt
run() { timeout 300 python bench.py --no-cpu-baseline --steps 48 --warmup 6 "$@" 2>&1 | tail -1 | \
  python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-60s %8.1f Mtriples/s  step %.3f ms  kernel %.3f ms  %6.0f GB/s  chunk %d' % (' '.join(sys.argv[1:]), d['value']/1e6, d['ms_per_step'], r['kernel_ms_avg'], r['achieved'], d['config']['triples_per_step_per_gpu']))" "$@"; }
run --workload netflix --dim 64 --sampler uniform
run --workload netflix --dim 64 --sampler adaptive
run --workload ml-20m --dim 128 --sampler uniform
run --workload ml-20m --dim 128 --sampler adaptive
run --workload msd --dim 256 --sampler uniform
run --workload msd --dim 256 --sampler adaptive
run --workload yelp --dim 128 --sampler uniform
run --workload yelp --dim 128 --sampler adaptive
run --workload ml-20m --dim 32 --sampler adaptive
run --workload ml-20m --dim 64 --sampler adaptive
run --workload ml-20m --dim 256 --sampler adaptive
run --workload ml-20m --dim 512 --sampler adaptive
