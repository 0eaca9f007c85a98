cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06n; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_vstream.py -q -x > $O/tests.log 2>&1; echo "rc=$?"; tail -3 $O/tests.log
bench() { timeout 300 python bench.py --workload yelp --no-cpu-baseline --steady-epochs 0 --sustained-epochs 3 $2 > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
try:
    j = json.loads(open("$O/$1.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("%-14s %.1f M triples/s  kernel %.4f ms  frac %.3f" % ("$1", j["value"] / 1e6, r["kernel_ms_avg"], r["frac"]))
except Exception as ex: print("$1 failed", ex)
PY
}
bench nobias ""; bench bias "--item-bias 1"; bench nobias2 ""; bench bias2 "--item-bias 1"
