#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_multirank_parity.py tests/test_gpu_sampler_stats.py -x -q -m gpu -k "not eight and not two_rank" 2>&1 | grep -E "passed|failed|^E " | head -5
python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | tail -1
