#!/bin/bash
# bench.py over the optimizer / workload / sampler matrix of the batched STREAM kernel (one line each)
for o in "$@"; do
  python bench.py --steps 30 --warmup 4 --no-cpu-baseline $o > gpurun_out/b.json 2>gpurun_out/b.err || tail -5 gpurun_out/b.err
  python - "$o" <<'PY'
import json, sys
j = json.load(open("gpurun_out/b.json")); r = j["roofline"]
print("%-55s | %7.1f M/s  ms/step %7.3f  kernel %7.3f ms  frac %.3f  loss %.4f" % (
    sys.argv[1], j["value"] / 1e6, j["ms_per_step"], r["kernel_ms_avg"], r["frac"], j["config"]["mean_bpr_loss"]))
PY
done
