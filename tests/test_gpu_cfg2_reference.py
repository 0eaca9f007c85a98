"""BASELINE configs[1] against the REFERENCE ITSELF (VERDICT r5 "missing" 2 / next 3 iv): tests/golden/
e2e_cfg2_reference.json holds the curves of the reference's own training loop (imported in place by
tests/golden/make_golden_cfg2.py in the build container) with the shape and semantics of configs/RQ1/ours.yaml.j2 —
Netflix-shaped 9,949 x 4,825, d = 64, mini-batches of 16, the same epoch order every epoch (`shuffle: false`),
UniformSampler, `item_bias`, SGD lr 0.05, evaluation by `RocAucOne` on one held-out positive per user against every
item outside the user's seen set — the regime SURVEY H1 flags: 10 k users and a batch of 16.  Twelve sampler seeds,
ten epochs.  Here, on the same data and initial tables:

  * STRICT — the reference's mini-batches of 16 through the library, replaying the reference's order every epoch
             (our Philox sampler instead of torch's generator): every epoch of the curve, all three metrics, raw
             +-0.002 + 2 se;
  * STREAM — the uniform throughput path as the product configures itself (`refresh_lag="auto"`, `launch_split="auto"`:
             lr 0.05 x 2 x 40,928 is outside the one-rank budget, so a period runs as two launches of 20,464 triples —
             2 per user row in flight against mini-batches of 16), its own device shuffle: same gate.  (One launch
             per period trails the reference's take-off at epochs 4 - 6 by 0.1 epoch — Recall@20 -0.0087 / -0.0075 —
             and meets it again from epoch 8 on: profiles/r06_cfg2.md.)
"""
import json
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

SEEDS = tuple(range(1, 13))


@pytest.fixture(scope="module")
def setting(golden_dir):
    from revisit_bpr.datasets import synthetic

    fix = json.loads((golden_dir / "e2e_cfg2_reference.json").read_text())
    cfg = fix["config"]
    data = synthetic.leave_one_out(synthetic.generate_latent(users=9_949, items=4_825, actions=563_577, factors=8,
                                                             strength=1.5, median_per_user=27, min_per_user=5, seed=7,
                                                             eval_users=0), 13)
    assert data.nnz == cfg["train_triples"] and len(data.eval_users) == cfg["eval_users"]
    dev = torch.device("cuda")
    t = {k: torch.from_numpy(getattr(data, k)).to(dev) for k in ("users", "items", "indptr", "indices", "eval_users",
                                                                 "eval_items")}
    # the TRAIN-only seen CSR (nDCG / Recall mask the training items, not the held-out positive)
    cnt = np.bincount(data.users, minlength=data.num_users)
    t["train_indptr"] = torch.from_numpy(np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)).to(dev)
    t["train_indices"] = t["items"]  # (training triples are sorted by user, then item)
    t["eval_indptr"] = torch.arange(len(data.eval_users) + 1, device=dev)
    return fix, data, t


def fresh_model(data, cfg):
    from revisit_bpr.models import BPR
    from revisit_bpr.models.bpr import MF

    torch.manual_seed(cfg["init_seed"])
    return BPR(fuse_forward=True, reg_alphas=cfg["reg"],
               logits_model=MF(torch.nn.Embedding(data.num_users, cfg["d"], padding_idx=0),
                               torch.nn.Embedding(data.num_items, cfg["d"], padding_idx=0), item_bias=True)).cuda()


@torch.no_grad()
def metrics(model, t):
    """RocAucOne (the product's class, fed whole blocks: column 0 = the positive, the rest every item, masked to the
    items outside the seen set — experiments/bpr/dataset.py:197-217 of the reference builds the same row per user),
    nDCG@100 / Recall@20 of the same positive with the TRAIN items masked."""
    from revisit_bpr.evaluation import evaluate_topk
    from revisit_bpr.metrics import RocAucOne

    model.eval()
    f = model.logits_model.get_features()
    P, Q, b = f["user"].data, f["item"].data, f["item_bias"].data
    auc = RocAucOne()
    auc.reset()
    I = Q.shape[0]
    eu, pos = t["eval_users"].long(), t["eval_items"].long()
    for lo in range(0, eu.numel(), 2048):
        u, p = eu[lo:lo + 2048], pos[lo:lo + 2048]
        logits = P[u] @ Q.T + b
        unseen = torch.ones(len(u), I, device=P.device)
        unseen[:, 0] = 0
        s_lo, s_hi = t["indptr"][u], t["indptr"][u + 1]
        cnt = s_hi - s_lo
        rows = torch.repeat_interleave(torch.arange(len(u), device=P.device), cnt)
        offs = torch.arange(int(cnt.sum()), device=P.device) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt)
        unseen[rows, t["indices"][torch.repeat_interleave(s_lo, cnt) + offs].long()] = 0
        out = torch.cat((logits.gather(1, p.unsqueeze(1)), logits), dim=1)
        mask = torch.cat((torch.ones(len(u), 1, device=P.device), unseen), dim=1)
        tgt = torch.zeros_like(out)
        tgt[:, 0] = 1.0
        auc(out, tgt, mask)
    top = evaluate_topk(P, Q, b, t["eval_users"], t["eval_indptr"], t["eval_items"], t["train_indptr"],
                        t["train_indices"], ks=(20, 100))
    model.train()
    return float(auc.get_metric()), top["ndcg@100"], top["recall@20"]


def compare(label, fix, ours, epochs):
    ref = fix["runs"]
    lines, ok = [], True
    for ep in epochs:
        for k, key in enumerate(("auc", "ndcg@100", "recall@20")):
            r = np.array([run[key][ep] for run in ref.values()])
            o = np.array([c[ep][k] for c in ours.values()])
            se = math.sqrt(r.var(ddof=1) / len(r) + o.var(ddof=1) / len(o))
            diff, tol = o.mean() - r.mean(), 0.002 + 2 * se
            lines.append(f"{label} {key} epoch {ep}: ours {o.mean():.4f}+-{o.std(ddof=1):.4f} (n={len(o)}) reference "
                         f"{r.mean():.4f}+-{r.std(ddof=1):.4f} (n={len(r)}) diff {diff:+.4f} tol {tol:.4f}")
            ok &= abs(diff) <= tol
    print("\n".join(lines))
    assert ok, "\n".join(lines)


def test_untrained_metrics_equal_the_reference(setting):
    fix, data, t = setting
    m = metrics(fresh_model(data, fix["config"]), t)
    for run in fix["runs"].values():
        for k, key in enumerate(("auc", "ndcg@100", "recall@20")):
            assert abs(m[k] - run[key][0]) < 2e-6, (key, m[k], run[key][0])


def test_strict_matches_the_reference_loop_at_cfg2(setting):
    from revisit_bpr import engine as eng

    fix, data, t = setting
    cfg = fix["config"]
    perm = torch.from_numpy(np.random.default_rng(cfg["order_seed"]).permutation(data.nnz)).cuda()
    users, items = t["users"][perm].contiguous(), t["items"][perm].contiguous()  # the same order every epoch
    ours = {}
    for seed in SEEDS:
        model = fresh_model(data, cfg)
        opt = torch.optim.SGD(model.parameters(), lr=cfg["lr"])
        model.bind_seen_csr(t["indptr"], t["indices"])
        sc = torch.zeros(4, device="cuda")
        curve = {0: metrics(model, t)}
        for ep in range(1, cfg["epochs"] + 1):
            model.train_strict(opt, users, items, cfg["B"], eng.NEG_UNIFORM, seed=seed, offset=(ep - 1) * data.nnz,
                               scalars=sc)
            curve[ep] = metrics(model, t)
        assert int(sc[3]) == cfg["epochs"] * data.nnz
        ours[seed] = curve
    compare("STRICT[B=16]", fix, ours, range(1, cfg["epochs"] + 1))


def test_stream_matches_the_reference_loop_at_cfg2(setting):
    from revisit_bpr.fast import StreamTrainer

    fix, data, t = setting
    cfg = fix["config"]
    ours = {}
    for seed in SEEDS:
        model = fresh_model(data, cfg)
        tr = StreamTrainer(model, t["users"], t["items"], t["indptr"], t["indices"], lr=cfg["lr"], sampler="uniform",
                           batch_size=cfg["B"], seed=seed, refresh_lag="auto")
        curve = {0: metrics(model, t)}
        for ep in range(1, cfg["epochs"] + 1):
            stats = tr.train_epoch()
            assert stats["triples"] == data.nnz
            curve[ep] = metrics(model, t)
        ours[seed] = curve
    assert tr.launch_split == 2 and tr.hot_lds == 0
    compare("STREAM[uniform, launches of %d]" % tr.chunk, fix, ours, range(1, cfg["epochs"] + 1))
