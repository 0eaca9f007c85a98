#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_h; mkdir -p $O
( time timeout 2400 python -m pytest tests/test_gpu_multirank_parity.py -x -q -m gpu --durations=15 ) > $O/pytest_multirank.txt 2>&1; tail -30 $O/pytest_multirank.txt
