#!/usr/bin/env python
"""Which start makes a half-epoch prefix at the ML-20M shape informative?  STRICT (the reference's
mini-batches) from (a) the reference init scaled by s, (b) the generator's latent factors x a in the first
16 columns; nDCG@100 / Recall@20 after 0 / 12 / 24 refresh periods.  GPU box."""
import math, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "revisit-bpr_amd")]
import numpy as np, torch
from revisit_bpr.datasets import synthetic
from revisit_bpr.models import BPR
from revisit_bpr.models.bpr import MF
from revisit_bpr import engine as eng
from revisit_bpr.evaluation import evaluate_topk

GEN = dict(users=136677, items=20108, actions=9_700_000, factors=16, strength=1.2, median_per_user=37,
           min_per_user=5, seed=13, eval_users=10_000, item_skew=1.2, item_shift=60.0)
D, B, P_GEO = 128, 256, 0.01
REG = {"user": 0.0016, "item": 0.0001, "neg": 0.00375}
data = synthetic.generate_latent(cache_dir=tempfile.gettempdir(), **GEN)
dev = torch.device("cuda")
t = {k: torch.from_numpy(getattr(data, k)).to(dev) for k in ("users", "items", "indptr", "indices", "eval_users", "eval_indptr", "eval_items")}
every = int(data.num_items * math.log(data.num_items) / B)
perm = torch.from_numpy(np.random.default_rng(13).permutation(data.nnz)).to(dev)
Z, Y = synthetic.latent_factors(GEN["users"], GEN["items"], GEN["factors"], GEN["seed"])

def metrics(model):
    model.eval()
    f = model.logits_model.get_features()
    out = evaluate_topk(f["user"].data, f["item"].data, None, t["eval_users"], t["eval_indptr"], t["eval_items"], t["indptr"], t["indices"], ks=(20, 100))
    model.train()
    return out["ndcg@100"], out["recall@20"]

def run(label, lr, init):
    torch.manual_seed(13)
    model = BPR(fuse_forward=True, reg_alphas=REG, logits_model=MF(torch.nn.Embedding(data.num_users, D, padding_idx=0), torch.nn.Embedding(data.num_items, D, padding_idx=0))).cuda()
    f = model.logits_model.get_features()
    init(f["user"].data, f["item"].data)
    opt = torch.optim.SGD(model.parameters(), lr=lr)
    model.bind_seen_csr(t["indptr"], t["indices"])
    model.engine().adaptive_refresh()
    out = [metrics(model)]
    sc = torch.zeros(4, device=dev)
    for k, periods in enumerate((12, 24, 47)):
        lo, hi = (0, 12, 24)[k] * every * B, min(periods * every * B, data.nnz)
        idx = perm[lo:hi]
        model.train_strict(opt, t["users"][idx].contiguous(), t["items"][idx].contiguous(), B, eng.NEG_ADAPTIVE, adaptive_p=P_GEO, seed=1, offset=lo, refresh_every=every, scalars=sc)
        out.append(metrics(model))
    print(label, "lr", lr, " ".join(f"{a:.4f}/{b:.4f}" for a, b in out), flush=True)

def scaled(s):
    def f(P, Q):
        P.mul_(s); Q.mul_(s)
    return f
def latent(a):
    def f(P, Q):
        P[:, :16] = torch.from_numpy(Z).to(dev) * a
        Q[:, :16] = torch.from_numpy(Y).to(dev) * a
    return f
for lr in (0.05,):
    run("init x1", lr, scaled(1.0))
    run("init x8", lr, scaled(8.0))
    run("init x32", lr, scaled(32.0))
    run("latent x0.5", lr, latent(0.5))
    run("latent x1", lr, latent(1.0))
    run("latent x2", lr, latent(2.0))
