#!/bin/bash
# r6 batch e: tests after the fixes, async cut with fold + LDS, cfg2 with launch_split auto, trainer path, k_vstream code size
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06e; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_hotlds.py tests/test_gpu_two_tier.py tests/test_gpu_config_run.py tests/test_gpu_bench.py tests/test_gpu_api.py "tests/test_gpu_parity.py::test_plan_epoch_sorted_input_promise_gives_the_same_plan" "tests/test_gpu_parity.py::test_async_cut_reads_the_table_whole_and_folds_on_demand" -q > $O/tests.log 2>&1
echo "tests rc=$?"; tail -8 $O/tests.log
timeout 1200 python -m pytest tests/test_gpu_cfg2_reference.py -q -s > $O/cfg2.log 2>&1
echo "cfg2 rc=$?"; grep -E "^STREAM.*epoch (3|4|5|6|7|10):|passed|failed" $O/cfg2.log | awk '{print $1,$2,$3,$4,$5,$6,$7,$(NF-3),$(NF-2),$(NF-1),$NF}' | sort -u | tail -24
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
run acut --async-cut 1
run acut_again --async-cut 1
run acut_cus64 --async-cut 1 --refresh-cus 64
run msd_acut --workload msd --steady-epochs 10 --steady-timed-epochs 10 --async-cut 1
timeout 300 python tools/eval_probe.py > $O/eval_probe.txt 2>&1; grep -E "evaluate_topk|auc" $O/eval_probe.txt | tail -8
EVAL_USERS=10000 LR=0.001 FULL_METRICS=1 timeout 900 python tools/bench_trainer_path.py ml-20m 128 > $O/trainer_path.txt 2>&1; tail -3 $O/trainer_path.txt
LR=0.001 EPOCHS=40,80,120,160 NSEEDS=8 timeout 900 python tools/fullepoch_study.py timed_lds_acut > $O/parity_acut_lr0.001.txt 2>&1; tail -2 $O/parity_acut_lr0.001.txt
bash profiles/r06_scripts/r06_vstream_codesize.sh 2>&1 | tail -12
