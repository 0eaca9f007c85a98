#!/bin/bash
# r6: the N > 1 code path with the LDS tier at full scale on ONE GPU (no node): emulated ranks, forced RCCL one rank
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06m; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_two_tier.py tests/test_gpu_multirank_parity.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
show() { python - <<PY
import json
try:
    j = json.loads(open("$O/$1.json").read().strip().splitlines()[-1]); r, e = j["roofline"], j.get("early_state", {})
    print("%-18s value %.1f M (step %.4f ms, kernel %.4f) early %.1f M lds rows %s | %s" % ("$1", j["value"] / 1e6, j["ms_per_step"], r["kernel_ms_avg"], e.get("value", 0) / 1e6, j["config"]["hot_lds"]["rows_in_lds_last_launch"], str(j["config"]["cadence"])[:110]))
    if "item_sync" in j: print("    item_sync", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in j["item_sync"].items() if not isinstance(v, dict)})
except Exception as ex: print("$1 parse failed", ex)
PY
}
timeout 600 python bench.py --no-cpu-baseline --emulate-ranks 8 --steady-timed-epochs 20 > $O/emu8.json 2> $O/emu8.err; show emu8
timeout 600 python bench.py --no-cpu-baseline --emulate-ranks 8 --steady-timed-epochs 20 --hot-lds 0 > $O/emu8_nolds.json 2> $O/emu8_nolds.err; show emu8_nolds
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29701 bench.py --gpus 1 --force-dist --no-cpu-baseline --steady-timed-epochs 20 > $O/forced.json 2> $O/forced.err; show forced
