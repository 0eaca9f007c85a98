#!/bin/bash
# A/B: walk vectors in flight per round (product = 2, tools/ubench/libbprcore_trips1.so = 1, r4's walk)
for lib in "" "$PWD/tools/ubench/libbprcore_trips1.so"; do
  echo "== ${lib:-product (2 trips)}"
  BPR_LIB_PATH=$lib python tools/trained_state_probe.py --lr 0.05 --marks 1,4 2>&1 | grep -A1 epoch | cut -c1-70
  BPR_LIB_PATH=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab.json 2>gpurun_out/ab.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print("bench: value %.1f M  region %.4f ms/step  kernel %.4f  steady_state %.1f M (%.4f ms/step, kernel %.4f)" % (j["value"]/1e6, j["timed_region"]["ms_per_step_measured"], j["roofline"]["kernel_ms_avg"], j["steady_state"]["value"]/1e6, j["steady_state"]["ms_per_step"], j["steady_state"]["kernel_ms_avg"]))
PY
done
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
