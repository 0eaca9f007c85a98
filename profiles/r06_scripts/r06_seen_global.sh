#!/bin/bash
# r6: every user's seen bitmap in HBM (SEEN_GLOBAL) -> no LDS bitmaps -> the CU's LDS holds twice the hot rows
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06g; mkdir -p $O
BPR_SEEN=global timeout 900 python -m pytest tests/test_gpu_hotlds.py -q > $O/tests_global.log 2>&1; tail -3 $O/tests_global.log
run() { # name, env..., -- args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 400 python bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    j = json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
    r, e = j["roofline"], j.get("early_state", {})
    print("%-22s value %.1f M (step %.4f ms, kernel %.4f, frac %.3f) early %.1f M (kernel %.4f) lds rows %s" % ("$name", j["value"] / 1e6, j["ms_per_step"], r["kernel_ms_avg"], r["frac"], e.get("value", 0) / 1e6, e.get("kernel_ms_avg", 0), j["config"]["hot_lds"]["rows_in_lds_last_launch"]))
except Exception as ex:
    print("$name parse failed", ex)
PY
}
run base X=1 --
run global_h256 BPR_SEEN=global --
run global_h320 BPR_SEEN=global -- --hot-rows 320
run global_h512 BPR_SEEN=global -- --hot-rows 512
run bitmap_h148 X=1 -- --hot-rows 148
run msd_base X=1 -- --workload msd --steady-epochs 10 --steady-timed-epochs 10
run msd_global BPR_SEEN=global -- --workload msd --steady-epochs 10 --steady-timed-epochs 10
run msd_global_h192 BPR_SEEN=global -- --workload msd --steady-epochs 10 --steady-timed-epochs 10 --hot-rows 192
