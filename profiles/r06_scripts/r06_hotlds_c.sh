#!/bin/bash
# r6: LDS tier — short runs at the end of a persistent workgroup's share (guided self-scheduling), lds_only, staggered flush
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hotlds.py -x -q > $O/tests_tail0.log 2>&1; tail -2 $O/tests_tail0.log
BPR_LDS_TAIL=25 timeout 600 python -m pytest tests/test_gpu_hotlds.py -x -q > $O/tests_tail25.log 2>&1; tail -2 $O/tests_tail25.log
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline ${EXTRA} > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    j = json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
    r, s = j["roofline"], j.get("steady_state", {})
    print("%-24s value %.1f M (kernel %.4f ms, step %.4f)  steady %.1f M (kernel %.4f ms, step %.4f)" % ("$name", j["value"] / 1e6, r["kernel_ms_avg"], j["ms_per_step"], s.get("value", 0) / 1e6, s.get("kernel_ms_avg", 0), s.get("ms_per_step", 0)))
except Exception as ex:
    print("$name parse failed", ex)
PY
}
run base BPR_HOT_LDS=0
for T in 0 8 15 25 35 50; do run lds_tail$T BPR_HOT_LDS=512 BPR_LDS_TAIL=$T; done
EXTRA="--item-bias 1" run bias_lds_tail15 BPR_HOT_LDS=512 BPR_LDS_TAIL=15
