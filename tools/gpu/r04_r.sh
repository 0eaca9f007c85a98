#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_r; mkdir -p $O
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 300 python -m pytest tests/test_gpu_vstream.py -x -q -m gpu -k "conserve" 2>&1 | tail -1; done | sort | uniq -c
timeout 600 python -m pytest tests/test_gpu_vstream.py -x -q -m gpu 2>&1 | tail -1
