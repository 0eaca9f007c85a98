#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_d; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_multirank_parity.py tests/test_gpu_bench.py -x -q -m gpu -k "eight or bench or check_rccl" > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt; grep "8 ranks vs 1\|8 processes" $O/pytest.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "mismatches" >> $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
bash tools/gpu/r04_emu.sh 2>&1 | tee $O/emu.txt
