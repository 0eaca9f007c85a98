#!/bin/bash
# does the step time depend on how long the job has run? (sustained epochs sweep, adaptive vs uniform)
for smp in adaptive uniform; do
  for e in 3 12 50; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sampler $smp --sustained-epochs $e > gpurun_out/sp.json 2>gpurun_out/sp.err
    python - $smp $e <<PY
import json,sys
j=json.loads(open("gpurun_out/sp.json").read().strip().splitlines()[-1])
print("%s epochs %s: region %.4f ms/step  sustained %.4f ms/step  kernel avg %.4f  loss %.4f" % (sys.argv[1], sys.argv[2], j["timed_region"]["ms_per_step_measured"], j["sustained"]["ms_per_step"], j["roofline"]["kernel_ms_avg"], j["config"]["mean_bpr_loss"]))
PY
  done
done
/opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | head -20
