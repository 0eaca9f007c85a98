"""Row 8e parity: user-sharded training with the replicated item table reconciled by ItemSync,
against the reference's SINGLE-process curves on the e2e parity set.  Two ranks share cuda:0 over
gloo here (RCCL needs one device per rank; the protocol — shards, chunk = refresh period / world,
asynchronous delta all-reduce — is identical).  Tolerance as in test_gpu_e2e_parity.py."""
import json
import math
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("kind,mode", [("uniform", "stream"), ("adaptive", "stream"),
                                       ("adaptive", "stream-lag"), ("adaptive", "stream-shard")])
def test_two_rank_training_matches_reference_curves(golden_dir, kind, mode):
    """mode stream-lag: the overlapped snapshot schedule (sort beside the previous launch, CU-masked
    streams) with the item reconciliation between the launches; stream-shard: every rank sorts half
    of the snapshot's factors, an all-gather shares them."""
    env = dict(os.environ, BPR_DIST_BACKEND="gloo")
    port = {"uniform": "29631", "adaptive": "29632"}[kind] if mode == "stream" else \
        {"stream-lag": "29634", "stream-shard": "29636"}[mode]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", port, str(ROOT / "tools" / "parity_multi.py"),
           kind, ",".join(str(s) for s in range(1, 31)), mode]  # 30 seeds: 5 made a 2-sigma gate flaky
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    runs = [json.loads(line) for line in res.stdout.splitlines() if line.startswith("{")]
    assert len(runs) == 30 and all(r["world"] == 2 for r in runs)
    # the ranks shuffle on the device: the yardstick is the reference over epoch orders
    ref = json.loads((golden_dir / "e2e_reference_sgd_orders.json").read_text())
    report, ok = [], True
    # A rank sees the other ranks' item updates up to two chunks late (one reconciliation in flight,
    # DESIGN.md 7).  On this 96 k-triple set a chunk is 5 % of an epoch (at ML-20M: 1 %), and the
    # uniform-sampler curve still climbs 0.01 per epoch at the end: the gate allows that lag times
    # the reference's local slope on top of the plateau tolerance.
    d = np.load(golden_dir / "e2e_data.npz")
    cfg = json.loads((golden_dir / "e2e_reference.json").read_text())["config"]
    I, n = int(d["num_items"]), len(d["users"])
    period = max(1, int(I * math.log(I) / cfg["B"])) * cfg["B"]
    lag_epochs = 2 * (period / 2) / n
    for key in ("ndcg@100", "recall@20"):
        for epoch in (-2, -1):
            o = np.array([r[key][epoch] for r in runs])
            rr = np.array([v[key][epoch] for k, v in ref["runs"].items() if k.startswith(kind)])
            prev = np.array([v[key][epoch - 1] for k, v in ref["runs"].items() if k.startswith(kind)])
            se = math.sqrt(o.var(ddof=1) / len(o) + rr.var(ddof=1) / len(rr))
            tol = 0.002 + lag_epochs * abs(rr.mean() - prev.mean()) + 2 * se
            diff = o.mean() - rr.mean()
            report.append(f"2 ranks {kind} {key} epoch {epoch}: ours {o.mean():.4f}±{o.std(ddof=1):.4f} "
                          f"ref {rr.mean():.4f}±{rr.std(ddof=1):.4f} diff {diff:+.4f} tol {tol:.4f}")
            ok &= abs(diff) <= tol
    print("\n".join(report))
    assert ok, "\n".join(report)


def _run_parity_multi(world, args, port):
    env = dict(os.environ, BPR_DIST_BACKEND="gloo", BPR_ADAM_LR="0.0005")
    if world == 1:
        cmd = [sys.executable, str(ROOT / "tools" / "parity_multi.py"), *args]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", port, str(ROOT / "tools" / "parity_multi.py"),
               *args]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    return [json.loads(line) for line in res.stdout.splitlines() if line.startswith("{")]


def test_two_rank_adam_training_matches_single_process():
    """BASELINE config 5 (Adam path, several GPUs): STRICT mini-batches per user shard + ItemSync
    (StrictTrainer) against the same trainer on one rank — itself pinned to the reference's Adam
    by the golden step tests.  Adam's bias-correction warm-up depends on the number of steps taken,
    so on this 400-steps-per-epoch set the first epochs of the 2-rank run lag; the plateau must
    agree (at ML-20M scale the curves agree from the first epoch: profiles/adam_2ranks_fullscale_r01.txt)."""
    one = _run_parity_multi(1, ["adaptive", "1,2,3,4,5", "adam"], "0")
    two = _run_parity_multi(2, ["adaptive", "1,2,3,4,5", "adam"], "29633")
    assert len(one) == 5 and len(two) == 5 and all(r["world"] == 2 for r in two)
    report, ok = [], True
    for key in ("ndcg@100", "recall@20"):
        for epoch in (-2, -1):
            a = np.array([r[key][epoch] for r in one])
            b = np.array([r[key][epoch] for r in two])
            se = math.sqrt(a.var(ddof=1) / len(a) + b.var(ddof=1) / len(b))
            tol = 0.003 + 2 * se  # local-Adam vs single-process Adam: same plateau, not the same run
            report.append(f"adam 2 ranks vs 1 {key} epoch {epoch}: {b.mean():.4f} vs {a.mean():.4f} "
                          f"diff {b.mean() - a.mean():+.4f} tol {tol:.4f}")
            ok &= abs(b.mean() - a.mean()) <= tol
    print("\n".join(report))
    assert ok, "\n".join(report)
    assert np.mean([r["ndcg@100"][-1] for r in two]) > 0.4  # it learned


def test_two_rank_batched_adam_matches_single_process():
    """BASELINE config 5 on its THROUGHPUT path: BatchedStreamTrainer (one launch per refresh
    period, virtual mini-batches, torch.optim.Adam semantics) on two user shards with the item table
    reconciled by ItemSync every period, against the same trainer on one rank (itself held to the
    reference's Adam curves by tests/test_gpu_e2e_parity.py).  Local Adam: the plateau must agree."""
    one = _run_parity_multi(1, ["adaptive", "1,2,3,4,5", "batched-adam"], "0")
    two = _run_parity_multi(2, ["adaptive", "1,2,3,4,5", "batched-adam"], "29635")
    assert len(one) == 5 and len(two) == 5 and all(r["world"] == 2 for r in two)
    report, ok = [], True
    for key in ("ndcg@100", "recall@20"):
        for epoch in (-2, -1):
            a = np.array([r[key][epoch] for r in one])
            b = np.array([r[key][epoch] for r in two])
            se = math.sqrt(a.var(ddof=1) / len(a) + b.var(ddof=1) / len(b))
            tol = 0.003 + 2 * se
            report.append(f"batched adam 2 ranks vs 1 {key} epoch {epoch}: {b.mean():.4f} vs {a.mean():.4f} "
                          f"diff {b.mean() - a.mean():+.4f} tol {tol:.4f}")
            ok &= abs(b.mean() - a.mean()) <= tol
    print("\n".join(report))
    assert ok, "\n".join(report)
    assert np.mean([r["ndcg@100"][-1] for r in two]) > 0.4


def test_item_sync_fused_step_equals_finish_then_start():
    """ItemSync.step() (one fused pass, bpr_item_fold_delta) == finish() + start() bit for bit."""
    import torch

    from revisit_bpr.distributed import ItemSync

    g = torch.Generator(device="cuda").manual_seed(5)
    Q0 = torch.randn(3001, 96, device="cuda", generator=g)
    Qa, Qb = Q0.clone(), Q0.clone()
    sa, sb = ItemSync([Qa], scale=0.5), ItemSync([Qb], scale=0.5)
    for k in range(4):
        upd = torch.randn(3001, 96, device="cuda", generator=g) * 0.01
        Qa += upd
        Qb += upd
        sa.finish()
        sa.start()
        sb.step()
    sa.finish()
    sb.finish()
    torch.cuda.synchronize()
    assert torch.equal(Qa, Qb) and torch.equal(sa.base[0], sb.base[0])
    assert not torch.equal(Qa, Q0)


def test_c_abi_comm_single_rank():
    """bpr_comm_init / bpr_item_sync (RCCL inside the library) with a communicator of one
    rank — all a 1-GPU box can run: librccl is found, the communicator is created, the delta /
    all-reduce / fold cycle leaves a lone replica exactly where training put it, the base follows
    it, and bpr_adaptive_refresh still sorts everything itself.  Against the torch twin
    (distributed.ItemSync) on the same updates: bit-identical tables."""
    import torch

    import oracle
    from revisit_bpr.distributed import ItemSync
    from revisit_bpr.engine import Engine

    g = torch.Generator(device="cuda").manual_seed(3)
    I, d = 3001, 64
    Q0 = torch.randn(I, d, device="cuda", generator=g) * 0.1
    Q0[0] = 0
    Qa, Qb = Q0.clone(), Q0.clone()
    bias = torch.zeros(I, device="cuda")
    e = Engine(torch.zeros(8, d, device="cuda"), Qa, bias)
    e.comm_init(Engine.comm_unique_id(), 0, 1)
    twin = ItemSync([Qb])
    for k in range(3):
        upd = torch.randn(I, d, device="cuda", generator=g) * 0.01
        Qa += upd
        Qb += upd
        bias += 0.01
        e.item_sync()
        twin.step()
    e.item_sync_finish()
    twin.finish()
    torch.cuda.synchronize()
    assert torch.equal(Qa, Qb) and not torch.equal(Qa, Q0)
    assert torch.allclose(bias, torch.full_like(bias, 0.03))
    e.adaptive_refresh()
    QT, _ = oracle.adaptive_stats(Qa.cpu().numpy())
    import numpy as np
    assert np.array_equal(e.adaptive_snapshot()[0].cpu().numpy(), oracle.adaptive_order(QT))
    e.comm_destroy()
    e.item_sync_finish if False else None
    with pytest.raises(Exception):
        e.item_sync()  # no communicator any more


def test_check_rccl_tool_single_rank():
    """tools/check_rccl.py — the one command for the first real node run — with the one rank a 1-GPU
    box has: ItemSync over backend "nccl", then the two-tier protocol through the C ABI
    (bpr_comm_init / bpr_comm_hot_tier / bpr_hot_sync / bpr_item_sync over RCCL) against the same
    protocol run in-process."""
    res = subprocess.run([sys.executable, str(ROOT / "tools" / "check_rccl.py")], capture_output=True, text=True,
                         timeout=600, env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    assert "OK two-tier" in res.stdout and "OK\n" in res.stdout


def test_eight_ranks_at_the_budget_cadence_match_one_rank():
    """The multi-rank gate at the tolerance north_star states: 8 ranks of the
    product trainer (StreamTrainer + ItemSync with the hot tier, the staleness-budget cadence, the
    snapshot schedule bench.py times) against ONE rank, full ML-20M shape, d = 128, lr 0.0094 (the
    reference's tuned SGD learning rate, ~10x the benchmark config's 0.001: the harder case), 20
    epochs, 24 seeds per side, ranks stepped in-process over distributed.LocalWorld.  The seed noise
    must be resolved (2 se <= 0.0018) and the difference must not be resolvably outside north_star's
    band: |diff| - 2 se <= 0.002 on nDCG@100 and Recall@20 at the last epoch; the raw numbers are
    printed.  (Five independent measurements of this point read -0.0001, +0.0019, +0.0022, +0.0009 and a
    pass: a raw gate at 0.002 with se 0.0007 trips on noise one time in ten.)
    profiles/r04_cadence_study.txt holds the sweep around this point."""
    base = [sys.executable, str(ROOT / "tools" / "cadence_study.py"), "--cadence", "auto", "--hot-rows", "1024",
            "--lr", "0.0094", "--epochs", "20", "--eval-every", "20"]
    # three processes beside each other on the one GPU (a run is launch-bound; the suite's wall clock)
    parts = [["--ranks", "1", "--seeds", "24"], ["--ranks", "8", "--seeds", "12"],
             ["--ranks", "8", "--seeds", "12", "--first-seed", "13"]]
    procs = [subprocess.Popen(base + p, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for p in parts]
    runs = []
    for proc in procs:
        out, err = proc.communicate(timeout=1500)
        assert proc.returncode == 0, err[-2000:]
        runs += [json.loads(line) for line in out.splitlines() if line.startswith("{")]
    one = [r for r in runs if r["world"] == 1]
    eight = [r for r in runs if r["world"] == 8]
    assert len(one) == 24 and sorted(r["seed"] for r in eight) == list(range(1, 25))
    assert max(r["replica_spread"] for r in eight) < 1e-4  # the replicas are one table after the epoch
    report, ok = [], True
    for key in ("ndcg@100", "recall@20"):
        a = np.array([r[key][-1] for r in one])
        b = np.array([r[key][-1] for r in eight])
        se = math.sqrt(a.var(ddof=1) / len(a) + b.var(ddof=1) / len(b))
        diff = b.mean() - a.mean()
        report.append(f"8 ranks vs 1 {key}: {b.mean():.4f} vs {a.mean():.4f} diff {diff:+.4f} (2 se {2 * se:.4f})")
        ok &= abs(diff) - 2 * se <= 0.002 and 2 * se <= 0.0018
    print("\n".join(report))
    assert ok, "\n".join(report)
    assert np.mean([r["ndcg@100"][-1] for r in eight]) > 0.4


def test_eight_rank_job_over_gloo():
    """The same job as EIGHT processes over gloo (sharing cuda:0; RCCL needs a device per rank): the
    collectives of all ranks line up through epochs of uneven shards (hot-tier exchange after every
    launch, cold all-reduce and snapshot refresh per chunk), and the curve is the one-process curve
    (one seed: a plumbing check with a loose band — the statistical gate is the test above)."""
    env = dict(os.environ, BPR_DIST_BACKEND="gloo", BPR_CADENCE="auto", BPR_HOT_ROWS="1024", BPR_LR="0.0094",
               BPR_EPOCHS="9")
    args = ["adaptive", "1", "stream-lag", "full"]
    one = subprocess.run([sys.executable, str(ROOT / "tools" / "parity_multi.py"), *args], env=env,
                         capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    many = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                           "--master-addr", "127.0.0.1", "--master-port", "29638",
                           str(ROOT / "tools" / "parity_multi.py"), *args], env=env, capture_output=True,
                          text=True, timeout=1500)
    assert many.returncode == 0, many.stderr[-2000:]
    a = [json.loads(line) for line in one.stdout.splitlines() if line.startswith("{")]
    b = [json.loads(line) for line in many.stdout.splitlines() if line.startswith("{")]
    assert len(a) == 1 and len(b) == 1 and all(r["world"] == 8 for r in b)
    da = np.mean([r["ndcg@100"][-1] for r in a])
    db = np.mean([r["ndcg@100"][-1] for r in b])
    print(f"8 processes over gloo vs 1: nDCG@100 after 9 epochs {db:.4f} vs {da:.4f}")
    # (epoch 9 of 20 is on the rising part of the curve, where 8 ranks trail one by up to 0.01:
    # profiles/r04_cadence_study.txt, dnDCG at epoch 10; the plateau is the gate above)
    assert da > 0.25 and abs(db - da) < 0.02, (da, db)
