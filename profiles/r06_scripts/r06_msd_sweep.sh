cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06h; mkdir -p $O
run() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 400 python bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    j = json.loads(open("$O/$name.json").read().strip().splitlines()[-1]); r, e = j["roofline"], j.get("early_state", {})
    print("%-22s value %.1f M (step %.4f ms, kernel %.4f, frac %.3f) early %.1f M (kernel %.4f) lds rows %s cus %s" % ("$name", j["value"] / 1e6, j["ms_per_step"], r["kernel_ms_avg"], r["frac"], e.get("value", 0) / 1e6, e.get("kernel_ms_avg", 0), j["config"]["hot_lds"]["rows_in_lds_last_launch"], j["config"]["refresh_schedule"]["side_stream_cus"]))
except Exception as ex: print("$name parse failed", ex)
PY
}
M="--workload msd --steady-epochs 10 --steady-timed-epochs 10"
run msd_b1024 X=1 -- $M
run msd_b768 BPR_LDS_BLOCK=768 -- $M
run msd_b512 BPR_LDS_BLOCK=512 -- $M
run msd_b1024_tail0 BPR_LDS_TAIL=0 -- $M
run msd_b1024_tail25 BPR_LDS_TAIL=25 -- $M
run msd_nolds X=1 -- $M --hot-lds 0
run msd_cus96 X=1 -- $M --refresh-cus 96
run yelp_sgd X=1 -- --workload yelp --optimizer sgd --steady-epochs 10 --steady-timed-epochs 10
