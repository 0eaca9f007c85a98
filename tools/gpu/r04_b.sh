#!/bin/bash
# r4 call 2: tail cut v2 (A/B + timeline), two-tier protocol tests, product metric tests on the device
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_two_tier.py tests/test_metrics_product.py -x -q -m gpu > $O/pytest_two_tier.txt 2>&1; echo "rc $?" >> $O/pytest_two_tier.txt
tail -15 $O/pytest_two_tier.txt
bash tools/gpu/r04_tail.sh
