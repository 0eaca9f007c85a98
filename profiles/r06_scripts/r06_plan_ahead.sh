cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06j; mkdir -p $O
run() { local name=$1; shift
  timeout 400 python bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    j = json.loads(open("$O/$name.json").read().strip().splitlines()[-1]); r, e = j["roofline"], j.get("early_state", {})
    print("%-22s value %.1f M (step %.4f ms, kernel %.4f) early %.1f M (kernel %.4f)" % ("$name", j["value"] / 1e6, j["ms_per_step"], r["kernel_ms_avg"], e.get("value", 0) / 1e6, e.get("kernel_ms_avg", 0)))
except Exception as ex: print("$name parse failed", ex); print(open("$O/$name.err").read()[-600:])
PY
}
run base
run plan_ahead_main --plan-ahead 2
run plan_ahead_side --plan-ahead 1
run base2
run plan_ahead_main2 --plan-ahead 2
