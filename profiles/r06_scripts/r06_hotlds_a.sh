#!/bin/bash
# r6: first contact of the LDS tier with the chip: its tests, then bench A/B (early + steady) with 0 / 64 / 128 LDS rows
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06a
timeout 900 python -m pytest tests/test_gpu_hotlds.py -x -q > gpurun_out/r06a/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r06a/tests.log
tail -15 gpurun_out/r06a/tests.log
for L in 0 128 64; do
  BPR_HOT_LDS=$L timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r06a/bench_lds$L.json 2> gpurun_out/r06a/bench_lds$L.err
  echo "lds=$L rc=$?"
  python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r06a/bench_lds$L.json").read().strip().splitlines()[-1])
    print("lds $L value", j["value"], "steady", j.get("steady_state", {}).get("value"), "roofline", j["roofline"].get("achieved"), j["roofline"].get("kernel_ms"))
except Exception as ex:
    print("parse failed", ex)
PY
done
