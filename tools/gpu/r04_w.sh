#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 1200 python -m pytest tests/test_gpu_vstream.py tests/test_gpu_bench.py tests/test_gpu_e2e_parity.py tests/test_gpu_fullsize.py tests/test_metrics_product.py -x -q -m gpu -k "not strict_optimizers and not strict_api and not cfg1" 2>&1 | grep -E "passed|failed|^E " | head -8
