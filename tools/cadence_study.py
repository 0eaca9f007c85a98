#!/usr/bin/env python
"""Multi-rank cadence study on ONE GPU: N ranks of the product trainer
(fast.StreamTrainer + distributed.ItemSync, user shards, replicated item table) stepped round-robin in
one process over `distributed.LocalWorld` — the same data flow as N processes, every sum folded one
protocol step after it was cut — on the full ML-20M-shaped set, d = 128.

    python tools/cadence_study.py --ranks 1,2,4,8 --lr 0.05 --epochs 4 --seeds 10 \
        --cadence rank --hot-rows 1024 --hot-split 1 [--cold-scale mean] [--lag 1]

One JSON line per run, then per configuration: mean nDCG@100 / Recall@20 per evaluated epoch, the
seed standard error and the difference to the 1-rank runs of the same invocation.
"""
import argparse
import json
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "revisit-bpr_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from revisit_bpr.datasets import synthetic  # noqa: E402
from revisit_bpr.distributed import ItemSync, LocalWorld, balanced_user_shards, owner_of  # noqa: E402
from revisit_bpr.evaluation import evaluate_topk  # noqa: E402
from revisit_bpr.fast import StreamTrainer  # noqa: E402
from revisit_bpr.models import BPR  # noqa: E402
from revisit_bpr.models.bpr import MF  # noqa: E402

REG = {"user": 0.0016, "item": 0.0001, "neg": 0.00375}


def load(which: str):
    if which == "full":
        return synthetic.generate_latent(136677, 20108, 9_700_000, factors=16, strength=1.2,
                                         median_per_user=37, min_per_user=5, seed=13,
                                         eval_users=10_000, item_skew=1.2, item_shift=60.0, cache_dir=tempfile.gettempdir()), 128
    d = np.load(ROOT / "tests/golden/e2e_data.npz")  # the 4,000-user parity set

    class D:
        pass

    o = D()
    for k in ("users", "items", "indptr", "indices", "eval_users", "eval_indptr", "eval_items"):
        setattr(o, k, d[k])
    o.num_users, o.num_items = int(d["num_users"]), int(d["num_items"])
    return o, 32


def run(data, dim, dev, t, world, seed, a, lr, epochs):
    bounds = balanced_user_shards(data.indptr, world)
    own = owner_of(data.users, bounds)
    U, I = data.num_users, data.num_items
    lw = LocalWorld(world)
    counts = torch.bincount(t["items"].long(), minlength=I)
    trs, models = [], []
    every = max(1, int(I * np.log(I) / 256))
    n_r = [int((own == r).sum()) for r in range(world)]
    from revisit_bpr import fast
    if a.budget is not None:
        fast.STALENESS_BUDGET = a.budget
    per_period = (world if a.cadence == "job" else 1 if a.cadence == "rank" else
                  fast.launches_per_period(lr, world, every * 256, fast.STALENESS_BUDGET))
    chunk = [max(1, min(every * 256 // per_period, n)) for n in n_r]
    rounds = max(-(-n // c) for n, c in zip(n_r, chunk))
    for r in range(world):
        torch.manual_seed(13)
        model = BPR(fuse_forward=True, reg_alphas=REG,
                    logits_model=MF(torch.nn.Embedding(U, dim, padding_idx=0),
                                    torch.nn.Embedding(I, dim, padding_idx=0))).to(dev)
        f = model.logits_model.get_features()
        mine = torch.from_numpy(own == r).to(dev)
        sync = None
        if world > 1:
            sync = ItemSync([f["item"].data], comm=lw.member(r), engine=model.engine(),
                            hot_rows=a.hot_rows if a.cadence != "job" or a.hot_job else 0,
                            item_counts=counts, scale=(1.0 / world if a.cold_scale == "mean" else 1.0))
        tr = StreamTrainer(model, t["users"][mine].contiguous(), t["items"][mine].contiguous(),
                           t["indptr"], t["indices"], lr=lr, sampler=a.sampler, adaptive_p=a.adaptive_p,
                           batch_size=256, seed=seed, rank=r, item_sync=sync, world=world,
                           cadence=a.cadence, hot_split=a.hot_split, rounds=rounds, sync_every=a.cold_every,
                           **({"refresh_lag": 1.0, "refresh_cus": 64} if a.lag else {}))
        trs.append(tr)
        models.append((model, f))
    curve = []
    for ep in range(epochs):
        for tr in trs:
            tr.epoch_begin()
        gens = [tr.epoch_iter() for tr in trs]
        alive = True
        while alive:
            alive = False
            for tr, g in zip(trs, gens):
                with tr.stream_scope():
                    try:
                        next(g)
                        alive = True
                    except StopIteration:
                        pass
        for tr in trs:
            tr.epoch_end()
        if (ep + 1) % a.eval_every == 0 or ep == epochs - 1:
            P = models[0][1]["user"].data.clone()
            for r in range(1, world):
                lo, hi = int(bounds[r]), int(bounds[r + 1])
                P[lo:hi] = models[r][1]["user"].data[lo:hi]
            m = evaluate_topk(P, models[0][1]["item"].data, None, t["eval_users"], t["eval_indptr"],
                              t["eval_items"], t["indptr"], t["indices"], ks=(20, 100))
            curve.append((ep + 1, m["ndcg@100"], m["recall@20"]))
    if world > 1:  # replicas agree after the last reconciliation (fp32 association aside)
        q0 = models[0][1]["item"].data
        spread = max(float((models[r][1]["item"].data - q0).abs().max()) for r in range(1, world))
    else:
        spread = 0.0
    for tr in trs:
        if tr.item_sync is not None:
            tr.item_sync.close()
    return curve, spread


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", default="full", choices=["full", "small"])
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--lr", type=float, default=0.05)
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--eval-every", type=int, default=1)
    ap.add_argument("--seeds", type=int, default=10)
    ap.add_argument("--first-seed", type=int, default=1, help="seeds first-seed .. first-seed + seeds - 1")
    ap.add_argument("--cadence", default="rank", choices=["rank", "job", "auto"])
    ap.add_argument("--budget", type=float, default=None, help="--cadence auto: override fast.STALENESS_BUDGET")
    ap.add_argument("--hot-rows", type=int, default=1024)
    ap.add_argument("--hot-job", action="store_true", help="hot tier also at the job cadence")
    ap.add_argument("--hot-split", type=int, default=1)
    ap.add_argument("--cold-every", type=int, default=1)
    ap.add_argument("--cold-scale", default="sum", choices=["sum", "mean"])
    ap.add_argument("--lag", type=int, default=1, help="1: the overlapped snapshot schedule bench.py times")
    ap.add_argument("--sampler", default="adaptive")
    ap.add_argument("--adaptive-p", type=float, default=0.01)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    data, dim = load(a.set)
    t = {k: torch.from_numpy(getattr(data, k)).to(dev) for k in
         ("users", "items", "indptr", "indices", "eval_users", "eval_indptr", "eval_items")}
    res = {}
    for world in [int(w) for w in a.ranks.split(",")]:
        for seed in range(a.first_seed, a.first_seed + a.seeds):
            t0 = time.time()
            curve, spread = run(data, dim, dev, t, world, seed, a, a.lr, a.epochs)
            res.setdefault(world, []).append(curve)
            print(json.dumps({"world": world, "seed": seed, "lr": a.lr, "cadence": a.cadence,
                              "hot_rows": a.hot_rows, "hot_split": a.hot_split, "cold_scale": a.cold_scale,
                              "cold_every": a.cold_every, "lag": a.lag, "epochs": [c[0] for c in curve],
                              "ndcg@100": [round(c[1], 5) for c in curve],
                              "recall@20": [round(c[2], 5) for c in curve],
                              "replica_spread": spread, "s": round(time.time() - t0, 1)}), flush=True)
    tag = (f"lr {a.lr:g} cadence {a.cadence} H {a.hot_rows} hot_split {a.hot_split} cold {a.cold_scale}"
           f"/every {a.cold_every} lag {a.lag}")
    base = None
    for world, runs in res.items():
        nd = np.array([[c[1] for c in r] for r in runs])
        rc = np.array([[c[2] for c in r] for r in runs])
        se = nd[:, -1].std(ddof=1) / np.sqrt(len(runs)) if len(runs) > 1 else float("nan")
        if world == 1:
            base = (nd.mean(0), rc.mean(0), se)
        line = (f"# {tag} | world {world} seeds {len(runs)} nDCG@100 {np.round(nd.mean(0), 4).tolist()} "
                f"(last se {se:.4f}) Recall@20 last {rc[:, -1].mean():.4f}")
        if base is not None and world != 1:
            dn, dr = nd.mean(0) - base[0], rc.mean(0) - base[1]
            line += (f" | vs 1 rank: dnDCG {np.round(dn, 4).tolist()} dRecall last {dr[-1]:+.4f} "
                     f"(2 se of the difference {2 * np.hypot(se, base[2]):.4f})")
        print(line, flush=True)


if __name__ == "__main__":
    main()
