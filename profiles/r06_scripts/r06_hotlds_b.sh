#!/bin/bash
# r6: LDS tier sweeps — workgroup size (rows that fit), hot-block replicas for the flush, hot rows in the global block
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06b; mkdir -p $O
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline ${EXTRA} > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    j = json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
    r = j["roofline"]
    print("%-28s value %.1f M steady %.1f M  kernel %.4f ms (early) " % ("$name", j["value"] / 1e6, j.get("steady_state", {}).get("value", 0) / 1e6, r.get("kernel_avg_ms", r.get("kernel_ms", 0)) or 0), j.get("steady_state", {}).get("kernel_avg_ms"))
except Exception as ex:
    print("$name parse failed", ex)
PY
}
run base BPR_HOT_LDS=0
run lds128_b1024 BPR_HOT_LDS=512
run lds_b768 BPR_HOT_LDS=512 BPR_LDS_BLOCK=768
run lds_b896 BPR_HOT_LDS=512 BPR_LDS_BLOCK=896
run lds_b640 BPR_HOT_LDS=512 BPR_LDS_BLOCK=640
run lds_b512 BPR_HOT_LDS=512 BPR_LDS_BLOCK=512
EXTRA="--hot-rows 256 --hot-replicas 4" run lds128_rep4 BPR_HOT_LDS=512
EXTRA="--hot-rows 256 --hot-replicas 8" run lds128_rep8 BPR_HOT_LDS=512
EXTRA="--hot-rows 128" run lds128_hot128 BPR_HOT_LDS=512
EXTRA="--hot-rows 1024" run lds128_hot1024 BPR_HOT_LDS=512
EXTRA="--item-bias 1" run bias_base BPR_HOT_LDS=0
EXTRA="--item-bias 1" run bias_lds BPR_HOT_LDS=512
