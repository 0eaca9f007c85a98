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
                          "--warmup", "2", "--cpu-seconds", "1", "--steady-timed-epochs", "3"],
                         capture_output=True, text=True, timeout=900)
    j = _line(res)
    assert j["n_gpus"] == 1 and j["unit"] == "triples/s" and j["value"] > 1e6 and j["dtype"] == "f32"
    assert j["vs_baseline"] is None and j["higher_is_better"] is True and j["data"] == "synthetic"
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic_measured_in_this_run"] is False
    assert j["roofline_atomic"]["line_atomics_per_triple"] >= 8.0
    assert j["cpu_baseline"]["cores"] >= 1 and j["cpu_baseline"]["value"] > 0
    sched = j["config"]["refresh_schedule"]
    assert sched["lag"] == 1.0 and sched["side_stream_cus"] >= 32  # the schedule the gates hold (32 CUs since the binned sort)
    sus = j["sustained"]  # whole epochs from random init, every plan in place
    st = j["steady_state"]  # whole epochs timed after 30 epochs of the same job: the trained state, the headline
    spe = j["config"]["steps_per_epoch"]
    assert st["epochs_trained_before"] >= 30 and st["steps"] == st["epochs"] * spe and st["value"] > 1e6
    assert st["first_epoch"]["epoch"] == st["epochs_trained_before"] + 1 and st["first_epoch"]["value"] > 1e6
    trained = (st["epochs_trained_before"] + st["epochs"]) * spe - ((6 + 2 + 1) // spe + 1 + sus["epochs"]) * spe
    assert j["config"]["triples_counted_by_kernel"] == (6 + sus["steps"] + trained) * j["config"]["triples_per_step_per_gpu"]
    assert sus["epochs"] >= 3 and sus["value"] > 1e6 and 0 < r["read_only_frac"] < r["frac"]
    # r6: the headline is the TRAINED state's whole-epoch wall-clock number, `steps` the steps behind it; the early
    # state and the driver's K-step region ride along, nothing modelled in either
    assert j["value"] == st["value"] and j["value_source"].startswith("steady state") and j["steps"] == st["steps"]
    assert j["early_state"]["value"] == sus["value"] and 0 < j["early_state"]["roofline_frac"] < 1
    assert r["kernel_ms_avg"] == st["kernel_ms_avg"] and j["ms_per_step"] == st["ms_per_step"]
    assert j["timed_region"]["steps"] == 6 and j["timed_region"]["value_measured"] > 1e6
    lds = j["config"]["hot_lds"]  # lr 0.001: inside the staleness budget -> asked for; a 5 % shard's launches may not fill the chip
    assert lds["rows_asked"] > 0 and lds["rows_in_lds_last_launch"] >= 0
    assert j["config"]["parity"]["tolerance_north_star"] == 0.002
    port = j["cpu_baseline_c_port"]  # the C restatement with OpenMP over triples
    assert port["kind"] == "port" and port["cores"] >= 1 and port["value"] > 0 and "OpenMP" in port["sample"]
    assert "optim.sgd.SGD" in j["cpu_baseline"]["sample"] or "SGD" in j["cpu_baseline"]["sample"]


def test_bench_workload_brings_its_own_hyperparameters():
    """--workload yelp = BASELINE configs[4]: Adam(0.1, 0.999), reg .0025/.0025/.00025, and a CPU baseline
    with the SAME optimizer (the committed cfg5 line carried an SGD baseline)."""
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "yelp", "--scale", "0.1",
                          "--steps", "4", "--warmup", "2", "--cpu-seconds", "1", "--sustained-epochs", "1",
                          "--steady-epochs", "0"],
                         capture_output=True, text=True, timeout=900)
    j = _line(res)
    w = j["config"]["workload"]
    assert "Adam lr=0.001 betas=(0.1, 0.999)" in w and "(0.0025, 0.0025, 0.00025)" in w and "d=128" in w
    assert "Adam" in j["cpu_baseline"]["sample"] and j["roofline"]["kernel"].startswith("k_vstream")


@pytest.mark.parametrize("cadence", ["auto", "job"])
def test_bench_forced_distributed_one_rank_through_rccl(cadence):
    """WORLD_SIZE = 1 under torch.distributed.run with backend nccl drives the exact
    N > 1 code path of bench.py through RCCL — process group on the device, two-tier ItemSync with both
    all-reduces on the side stream (auto: fused bpr_sync_cut; job: the sharded refresh's all-gather), the
    cold message as reduce-scatter + all-gather behind the size switch — before a node ever sees it."""
    env = dict(os.environ)
    env.pop("BPR_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29673", str(ROOT / "bench.py"), "--gpus", "1",
           "--force-dist", "--scale", "0.05", "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
           "--cadence", cadence, "--sustained-epochs", "1", "--steady-epochs", "0", "--split-cold-mb", "1"]
    j = _line(subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900))
    assert j["rccl_ranks_seen"] == 1 and j["dist_backend"] == "nccl" and j["forced_distributed"] is True
    assert j["item_sync"]["all_reduces"] >= 6 and j["item_sync"]["all_reduce_ms_avg"] > 0
    if cadence == "job":
        assert j["config"]["refresh_schedule"]["sharded_over_ranks"] is True
    else:
        assert j["item_sync"]["hot"]["all_reduces"] >= 6 and j["config"]["refresh_schedule"]["lag"] == 1.0
    steps = 6 + j["sustained"]["steps"]
    assert j["config"]["triples_counted_by_kernel"] == steps * j["config"]["triples_per_step_per_gpu"]


@pytest.mark.parametrize("cadence", ["auto", "job"])
def test_bench_two_ranks_over_gloo(cadence):
    env = dict(os.environ, BPR_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29671", str(ROOT / "bench.py"), "--gpus", "2",
           "--scale", "0.05", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--cadence", cadence,
           "--sustained-epochs", "1", "--steady-epochs", "0"]
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
