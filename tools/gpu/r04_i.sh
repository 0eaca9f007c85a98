#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_i; mkdir -p $O
( time timeout 2400 python -m pytest tests/test_gpu_multirank_parity.py -x -q -m gpu ) > $O/pytest_multirank.txt 2>&1; tail -12 $O/pytest_multirank.txt; grep "8 ranks vs 1\|8 processes" $O/pytest_multirank.txt
