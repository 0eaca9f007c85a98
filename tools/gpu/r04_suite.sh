#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( time timeout 3300 python -m pytest tests -q -m gpu --durations=120 ) > $O/pytest_gpu_r04.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu_r04.log
tail -12 $O/pytest_gpu_r04.log
python bench.py --steps 20 --warmup 5 > $O/bench_driverlike.log 2>&1; tail -1 $O/bench_driverlike.log > $O/bench_r04_driverlike.json; cut -c1-300 $O/bench_r04_driverlike.json
