"""The metric's shape against the REFERENCE ITSELF (VERDICT r4 "missing" 2): tests/golden/
e2e_ml20m_reference_prefix.json holds nDCG@100 / Recall@20 of the reference's own training loop
(/root/reference/example.py:157-192 + AdaptiveSampler, imported in place by
tests/golden/make_golden_fullscale.py in the build container) after 12, 24 (, 36, 47) refresh periods
of the first epoch on the ML-20M-shaped set of the full-scale gates (136,677 x 20,108, 9.6 M triples,
d = 128, SGD lr 0.05, B = 256, adaptive p = 1/100), three sampler seeds.  Here, on the same data, the
same initial tables and for the same number of triples:

  * STRICT  — the reference's mini-batches through the library, replaying the reference's epoch order
              (same batches, our Philox sampler instead of torch's generator);
  * STREAM  — the throughput path with the schedule bench.py times (snapshot one launch older, sorted
              beside the launch on 64 masked CUs) and with the reference's schedule; its own device
              shuffle, one launch per refresh period.

Gate: |difference of seed means| <= 0.002 (BASELINE.json) + 2 standard errors, at every checkpoint the
fixture holds; every number is printed.  The first epoch is the take-off of the curve (0.002 untrained ->
0.004 -> 0.03 -> ...), so the later checkpoints are the informative ones."""
import json
import math
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

SEEDS = (1, 2, 3, 4, 5, 6)


@pytest.fixture(scope="module")
def setting(golden_dir):
    from revisit_bpr.datasets import synthetic

    fix = json.loads((golden_dir / "e2e_ml20m_reference_prefix.json").read_text())
    cfg = fix["config"]
    data = synthetic.generate_latent(136677, 20108, 9_700_000, factors=16, strength=1.2, median_per_user=37,
                                     min_per_user=5, seed=13, eval_users=10_000, item_skew=1.2, item_shift=60.0,
                                     cache_dir=tempfile.gettempdir())
    assert data.nnz == cfg["train_triples"]  # the generator still draws the set the fixture was made on
    dev = torch.device("cuda")
    t = {k: torch.from_numpy(getattr(data, k)).to(dev)
         for k in ("users", "items", "indptr", "indices", "eval_users", "eval_indptr", "eval_items")}
    return fix, data, t


def fresh_model(data, cfg):
    from revisit_bpr.models import BPR
    from revisit_bpr.models.bpr import MF

    torch.manual_seed(cfg["init_seed"])
    return BPR(fuse_forward=True, reg_alphas=cfg["reg"],
               logits_model=MF(torch.nn.Embedding(data.num_users, cfg["d"], padding_idx=0),
                               torch.nn.Embedding(data.num_items, cfg["d"], padding_idx=0))).cuda()


def metrics(model, t):
    from revisit_bpr.evaluation import evaluate_topk

    model.eval()
    f = model.logits_model.get_features()
    out = evaluate_topk(f["user"].data, f["item"].data, None, t["eval_users"], t["eval_indptr"],
                        t["eval_items"], t["indptr"], t["indices"], ks=(20, 100))
    model.train()
    return out["ndcg@100"], out["recall@20"]


def compare(label, fix, ours):
    """ours: {seed: {periods: (ndcg, recall)}}"""
    ref = fix["runs"]
    lines, ok = [], True
    for periods in fix["config"]["checkpoint_periods"]:
        if periods == 0:
            continue
        for k, key in enumerate(("ndcg@100", "recall@20")):
            r = np.array([run[str(periods)][key] for run in ref.values() if str(periods) in run])
            o = np.array([c[periods][k] for c in ours.values()])
            se = math.sqrt(r.var(ddof=1) / len(r) + o.var(ddof=1) / len(o))
            diff, tol = o.mean() - r.mean(), 0.002 + 2 * se
            lines.append(f"{label} {key} after {periods} periods: ours {o.mean():.4f}+-{o.std(ddof=1):.4f} (n={len(o)}) "
                         f"reference {r.mean():.4f}+-{r.std(ddof=1):.4f} (n={len(r)}) diff {diff:+.4f} tol {tol:.4f}")
            ok &= abs(diff) <= tol
    print("\n".join(lines))
    assert ok, "\n".join(lines)


def test_untrained_metrics_equal_the_reference(setting):
    fix, data, t = setting
    nd, rc = metrics(fresh_model(data, fix["config"]), t)
    for run in fix["runs"].values():
        assert abs(nd - run["0"]["ndcg@100"]) < 2e-6 and abs(rc - run["0"]["recall@20"]) < 2e-6


def test_strict_matches_the_reference_loop_at_ml20m_shape(setting):
    from revisit_bpr import engine as eng

    fix, data, t = setting
    cfg = fix["config"]
    B, every = cfg["B"], cfg["refresh_every_batches"]
    perm = torch.from_numpy(np.random.default_rng(cfg["order_seed"]).permutation(data.nnz)).cuda()
    marks = [p for p in cfg["checkpoint_periods"] if p > 0]
    ours = {}
    for seed in SEEDS:
        model = fresh_model(data, cfg)
        opt = torch.optim.SGD(model.parameters(), lr=cfg["lr"])
        model.bind_seen_csr(t["indptr"], t["indices"])
        model.engine().adaptive_refresh()  # AdaptiveSampler.update_stats() before the loop (example.py:304)
        sc = torch.zeros(4, device="cuda")
        curve, lo = {}, 0
        for periods in marks:
            hi = periods * every * B
            idx = perm[lo:hi]
            model.train_strict(opt, t["users"][idx].contiguous(), t["items"][idx].contiguous(), B,
                               eng.NEG_ADAPTIVE, adaptive_p=cfg["adaptive_p"], seed=seed, offset=lo,
                               refresh_every=every, scalars=sc)
            lo = hi
            curve[periods] = metrics(model, t)
        assert int(sc[3]) == lo
        ours[seed] = curve
    compare("STRICT", fix, ours)


@pytest.mark.parametrize("schedule", ["timed", "reference"])
def test_stream_matches_the_reference_loop_at_ml20m_shape(setting, schedule):
    from revisit_bpr.fast import StreamTrainer

    fix, data, t = setting
    cfg = fix["config"]
    kw = {"timed": dict(refresh_lag=1.0, refresh_cus=64), "reference": {}}[schedule]
    marks = [p for p in cfg["checkpoint_periods"] if p > 0]
    ours = {}
    for seed in SEEDS:
        model = fresh_model(data, cfg)
        tr = StreamTrainer(model, t["users"], t["items"], t["indptr"], t["indices"], lr=cfg["lr"],
                           sampler="adaptive", adaptive_p=cfg["adaptive_p"], batch_size=cfg["B"], seed=seed, **kw)
        assert tr.chunk == cfg["refresh_every_batches"] * cfg["B"]  # a launch = a refresh period
        curve, done = {}, 0
        for periods in marks:
            stats = tr.train_chunks(periods - done)
            done = periods
            assert stats["triples"] == periods * tr.chunk
            curve[periods] = metrics(model, t)
        ours[seed] = curve
    compare(f"STREAM[{schedule} schedule]", fix, ours)
