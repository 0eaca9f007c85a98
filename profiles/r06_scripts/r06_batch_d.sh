#!/bin/bash
# r6 batch d: new tests (LDS tier, cfg2 vs the reference, 2-rank item_bias, fused eval, plan promise, bench line),
# the new headline, MSD / async cut with the tier, the trainer path, the evaluation's cost
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06d; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_hotlds.py tests/test_gpu_two_tier.py tests/test_gpu_config_run.py tests/test_gpu_bench.py "tests/test_gpu_parity.py::test_plan_epoch_sorted_input_promise_gives_the_same_plan" -x -q > $O/tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/tests.log
timeout 1200 python -m pytest tests/test_gpu_cfg2_reference.py -x -q -s > $O/cfg2.log 2>&1
echo "cfg2 rc=$?"; grep -E "epoch (1|5|10):|passed|failed" $O/cfg2.log | tail -24
run() { # name, args...
  local name=$1; shift
  timeout 400 python bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    j = json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
    r, s, e = j["roofline"], j.get("steady_state", {}), j.get("early_state", {})
    print("%-22s value %.1f M (step %.4f ms, kernel %.4f, frac %.3f) early %.1f M (kernel %.4f) lds rows %s plan %.3f ms" % ("$name", j["value"] / 1e6, j["ms_per_step"], r["kernel_ms_avg"], r["frac"], e.get("value", 0) / 1e6, e.get("kernel_ms_avg", 0), j["config"]["hot_lds"]["rows_in_lds_last_launch"], j["config"]["plan_epoch"]["ms"]))
except Exception as ex:
    print("$name parse failed", ex)
PY
}
run default
run default_again
run nolds --hot-lds 0
run acut --async-cut 1
run bias --item-bias 1
run msd --workload msd --steady-epochs 10 --steady-timed-epochs 10
run msd_nolds --workload msd --steady-epochs 10 --steady-timed-epochs 10 --hot-lds 0
run netflix --workload netflix --sampler adaptive --steady-timed-epochs 20
run d64 --dim 64 --steady-timed-epochs 30
timeout 300 python tools/eval_probe.py > $O/eval_probe.txt 2>&1; tail -14 $O/eval_probe.txt
EVAL_USERS=10000 LR=0.001 FULL_METRICS=1 timeout 900 python tools/bench_trainer_path.py ml-20m 128 > $O/trainer_path.txt 2>&1; tail -5 $O/trainer_path.txt
