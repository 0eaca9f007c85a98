#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_j; mkdir -p $O
bash tools/gpu/r04_vs_occ.sh
( time timeout 1200 python -m pytest tests/test_gpu_multirank_parity.py -x -q -m gpu --durations=5 -k "eight" ) > $O/pytest_eight.txt 2>&1; tail -15 $O/pytest_eight.txt
