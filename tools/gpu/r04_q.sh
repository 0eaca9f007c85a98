#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_q; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | tail -2
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vstream.py tests/test_gpu_two_tier.py tests/test_gpu_api.py -x -q -m gpu ) > $O/pytest_core.txt 2>&1; tail -5 $O/pytest_core.txt
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-260
