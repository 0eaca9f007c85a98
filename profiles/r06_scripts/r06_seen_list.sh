cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06k; mkdir -p $O
BPR_SEEN=list timeout 900 python -m pytest tests/test_gpu_hotlds.py -q > $O/tests_list.log 2>&1; tail -3 $O/tests_list.log
run() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 400 python bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    j = json.loads(open("$O/$name.json").read().strip().splitlines()[-1]); r, e = j["roofline"], j.get("early_state", {})
    print("%-22s value %.1f M (step %.4f ms, kernel %.4f, frac %.3f) early %.1f M (kernel %.4f) lds rows %s" % ("$name", j["value"] / 1e6, j["ms_per_step"], r["kernel_ms_avg"], r["frac"], e.get("value", 0) / 1e6, e.get("kernel_ms_avg", 0), j["config"]["hot_lds"]["rows_in_lds_last_launch"]))
except Exception as ex: print("$name parse failed", ex); print(open("$O/$name.err").read()[-500:])
PY
}
M="--workload msd --steady-epochs 10 --steady-timed-epochs 10"
run msd_bitmap X=1 -- $M
run msd_list BPR_SEEN=list -- $M
run msd_list_h160 BPR_SEEN=list -- $M --hot-rows 160
run ml20m_list BPR_SEEN=list -- --steady-timed-epochs 30
run ml20m_list_h256 BPR_SEEN=list -- --steady-timed-epochs 30 --hot-rows 320
run yelp_sgd_lds X=1 -- --workload yelp --optimizer sgd --steady-epochs 10 --steady-timed-epochs 10 --hot-lds 512 --refresh-lag 1 --launch-split 1
run yelp_sgd_nolds X=1 -- --workload yelp --optimizer sgd --steady-epochs 10 --steady-timed-epochs 10 --hot-lds 0 --refresh-lag 1 --launch-split 1
