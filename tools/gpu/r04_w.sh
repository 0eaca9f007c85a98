#!/bin/bash
cd "$(dirname "$0")/../.."
python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vstream.py -x -q -m gpu 2>&1 | grep -E "passed|failed|^E " | head -5
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/bench_r04_driverlike.json; cut -c1-330 gpurun_out/bench_r04_driverlike.json
