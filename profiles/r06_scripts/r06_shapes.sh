cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06i; mkdir -p $O
run() { local name=$1; shift
  timeout 400 python bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    j = json.loads(open("$O/$name.json").read().strip().splitlines()[-1]); r, e = j["roofline"], j.get("early_state", {})
    print("%-22s value %.1f M (step %.4f ms, kernel %.4f, frac %.3f) early %.1f M (kernel %.4f) lds rows %s cus %s" % ("$name", j["value"] / 1e6, j["ms_per_step"], r["kernel_ms_avg"], r["frac"], e.get("value", 0) / 1e6, e.get("kernel_ms_avg", 0), j["config"]["hot_lds"]["rows_in_lds_last_launch"], j["config"]["refresh_schedule"]["side_stream_cus"]))
except Exception as ex: print("$name parse failed", ex)
PY
}
run msd --workload msd --steady-epochs 10 --steady-timed-epochs 10
run yelp_sgd --workload yelp --optimizer sgd --steady-epochs 10 --steady-timed-epochs 10
run yelp_adam --workload yelp --steady-epochs 10 --steady-timed-epochs 10
run yelp_adam_bias --workload yelp --steady-epochs 10 --steady-timed-epochs 10 --item-bias 1
run netflix --workload netflix --steady-timed-epochs 20
run netflix_lsplit1 --workload netflix --steady-timed-epochs 20 --launch-split 1
run default
