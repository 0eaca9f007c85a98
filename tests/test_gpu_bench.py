"""bench.py end to end (small --scale): the single-GPU line the driver records, and the N > 1 path
(two ranks sharing cuda:0 over gloo — RCCL needs one device per rank; the protocol is the same:
the staleness-budget cadence with the two-tier reconciliation, and r3's job cadence with the
sharded snapshot refresh + all-gather)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _line(res):
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-1000:]
    return json.loads(lines[0])


def test_bench_single_gpu_line():
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--scale", "0.05", "--steps", "6",
                          "--warmup", "2", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=900)
    j = _line(res)
    assert j["n_gpus"] == 1 and j["unit"] == "triples/s" and j["value"] > 1e6 and j["dtype"] == "f32"
    assert j["vs_baseline"] is None and j["higher_is_better"] is True and j["data"] == "synthetic"
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic_measured_in_this_run"] is False
    assert j["roofline_atomic"]["line_atomics_per_triple"] >= 8.0
    assert j["cpu_baseline"]["cores"] >= 1 and j["cpu_baseline"]["value"] > 0
    sched = j["config"]["refresh_schedule"]
    assert sched["lag"] == 1.0 and sched["side_stream_cus"] >= 64  # the schedule the gates hold
    sus = j["sustained"]  # whole epochs, every plan in place
    assert j["config"]["triples_counted_by_kernel"] == (6 + sus["steps"]) * j["config"]["triples_per_step_per_gpu"]
    assert sus["epochs"] == 3 and sus["value"] > 1e6 and 0 < r["read_only_frac"] < r["frac"]
    assert j["config"]["parity"]["tolerance_north_star"] == 0.002


@pytest.mark.parametrize("cadence", ["auto", "job"])
def test_bench_two_ranks_over_gloo(cadence):
    env = dict(os.environ, BPR_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29671", str(ROOT / "bench.py"), "--gpus", "2",
           "--scale", "0.05", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--cadence", cadence,
           "--sustained-epochs", "1"]
    j = _line(subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900))
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 1e6
    assert j["config"]["cadence"].startswith(cadence)
    assert j["item_sync"]["all_reduces"] >= 6
    if cadence == "job":
        assert j["config"]["refresh_schedule"]["sharded_over_ranks"] is True
    else:  # lr 0.001: a full period per rank, hot tier exchanged after every launch
        assert "1 chunk(s)" in j["config"]["cadence"] and j["item_sync"]["hot"]["all_reduces"] >= 6
        assert j["config"]["refresh_schedule"]["lag"] == 1.0
    steps = 6 + j["sustained"]["steps"]
    assert j["config"]["triples_counted_by_kernel"] == steps * j["config"]["triples_per_step_per_gpu"]
