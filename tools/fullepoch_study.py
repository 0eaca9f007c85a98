#!/usr/bin/env python
"""Where does the STREAM path leave the reference's curve at the END of the first epoch?

tests/golden/e2e_ml20m_reference_prefix.json (r5: the reference's own loop run on to 36 and 47 refresh
periods) against variants of the STREAM schedule on the same data, same initial tables, same number of
triples; nDCG@100 / Recall@20 at the fixture's checkpoints, seed means.  MI355X only.

    python tools/fullepoch_study.py [variant ...]        (default: all)
    LR=0.01 EPOCHS=1,2,3,4,6 python tools/fullepoch_study.py strict timed reference
        another learning rate / checkpoints after whole epochs (no reference runs there: STRICT — pinned to
        the reference by the fixture at lr 0.05 — stands for it; its epochs repeat the fixture's order)
"""
import os
import json
import math
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "revisit-bpr_amd"))

from revisit_bpr import engine as eng  # noqa: E402
from revisit_bpr.datasets import synthetic  # noqa: E402
from revisit_bpr.evaluation import evaluate_topk  # noqa: E402
from revisit_bpr.fast import StreamTrainer  # noqa: E402
from revisit_bpr.models import BPR  # noqa: E402
from revisit_bpr.models.bpr import MF  # noqa: E402

SEEDS = tuple(range(1, 1 + int(os.environ.get("NSEEDS", "3"))))
VARIANTS = {
    "timed": dict(refresh_lag=1.0, refresh_cus=64),
    "reference": {},
    "lag1_unmasked": dict(refresh_lag=1.0),
    "ref_inflight4096": dict(max_inflight=4096),
    "ref_inflight2048": dict(max_inflight=2048),
    "ref_inflight1024": dict(max_inflight=1024),
    "ref_inflight512": dict(max_inflight=512),
    "ref_inflight256": dict(max_inflight=256),
    "ref_split2": dict(refresh_split=2),
    # r6: the LDS tier of the hot block, forced on whatever the learning rate ("auto" = what the trainer picks)
    "auto": dict(refresh_lag="auto"),
    "timed_lds": dict(refresh_lag=1.0, refresh_cus=-1, hot_lds=512),
    "timed_nolds": dict(refresh_lag=1.0, refresh_cus=-1, hot_lds=0),
    "timed_lds_acut": dict(refresh_lag=1.0, refresh_cus=-1, hot_lds=512, async_cut=True),
    "reference_lds": dict(hot_lds=512),
    "reference_nolds": dict(hot_lds=0),
    # r6: a period as 2 / 4 launches reading the same snapshot (a user's triples of a period no longer back to back)
    "ref_lsplit2": dict(launch_split=2, hot_lds=0),
    "ref_lsplit4": dict(launch_split=4, hot_lds=0),
}


def main():
    fix = json.loads((ROOT / "tests/golden/e2e_ml20m_reference_prefix.json").read_text())
    cfg = fix["config"]
    data = synthetic.generate_latent(136677, 20108, 9_700_000, factors=16, strength=1.2, median_per_user=37,
                                     min_per_user=5, seed=13, eval_users=10_000, item_skew=1.2, item_shift=60.0,
                                     cache_dir=tempfile.gettempdir())
    assert data.nnz == cfg["train_triples"]
    t = {k: torch.from_numpy(getattr(data, k)).cuda()
         for k in ("users", "items", "indptr", "indices", "eval_users", "eval_indptr", "eval_items")}
    marks = [p for p in cfg["checkpoint_periods"] if p > 0]
    extra = sorted(set(marks) | {40, 44})
    if "MARKS" in os.environ:
        extra = [int(x) for x in os.environ["MARKS"].split(",")]
    lr = float(os.environ.get("LR", cfg["lr"]))
    epochs = [int(x) for x in os.environ["EPOCHS"].split(",")] if "EPOCHS" in os.environ else None
    if epochs:  # whole epochs instead of refresh periods
        extra, marks = epochs, []
    if lr != cfg["lr"]:
        marks = []

    def fresh():
        torch.manual_seed(cfg["init_seed"])
        return BPR(fuse_forward=True, reg_alphas=cfg["reg"],
                   logits_model=MF(torch.nn.Embedding(data.num_users, cfg["d"], padding_idx=0),
                                   torch.nn.Embedding(data.num_items, cfg["d"], padding_idx=0))).cuda()

    def metrics(model):
        model.eval()
        f = model.logits_model.get_features()
        out = evaluate_topk(f["user"].data, f["item"].data, None, t["eval_users"], t["eval_indptr"],
                            t["eval_items"], t["indptr"], t["indices"], ks=(20, 100))
        model.train()
        return float(out["ndcg@100"]), float(out["recall@20"])

    ref = {p: np.array([[run[str(p)]["ndcg@100"], run[str(p)]["recall@20"]] for run in fix["runs"].values()
                        if str(p) in run]) for p in marks}
    print(f"lr {lr}", flush=True)
    print("reference: " + "  ".join(f"{p}: {ref[p][:, 0].mean():.4f}/{ref[p][:, 1].mean():.4f}" for p in marks), flush=True)
    names = sys.argv[1:] or list(VARIANTS)
    for name in names:
        if name == "strict":
            curves = []
            B, every = cfg["B"], cfg["refresh_every_batches"]
            perm = torch.from_numpy(np.random.default_rng(cfg["order_seed"]).permutation(data.nnz)).cuda()
            for seed in SEEDS:
                model = fresh()
                opt = torch.optim.SGD(model.parameters(), lr=lr)
                model.bind_seen_csr(t["indptr"], t["indices"])
                model.engine().adaptive_refresh()
                sc = torch.zeros(4, device="cuda")
                curve, lo = {}, 0
                for p in extra:
                    hi = p * every * B if not epochs else p * data.nnz
                    idx = perm[torch.arange(lo, hi, device="cuda") % data.nnz]  # (later epochs: the same order again)
                    model.train_strict(opt, t["users"][idx].contiguous(), t["items"][idx].contiguous(), B,
                                       eng.NEG_ADAPTIVE, adaptive_p=cfg["adaptive_p"], seed=seed, offset=lo,
                                       refresh_every=every, scalars=sc)
                    lo = hi
                    curve[p] = metrics(model)
                curves.append(curve)
        else:
            kw = VARIANTS[name]
            curves = []
            for seed in SEEDS:
                model = fresh()
                tr = StreamTrainer(model, t["users"], t["items"], t["indptr"], t["indices"], lr=lr,
                                   sampler="adaptive", adaptive_p=cfg["adaptive_p"], batch_size=cfg["B"], seed=seed, **kw)
                per = kw.get("refresh_split", 1) * kw.get("launch_split", 1)
                curve, done = {}, 0
                for p in extra:
                    if epochs:
                        for _ in range(p - done):
                            tr.train_epoch()
                    else:
                        tr.train_chunks((p - done) * per)
                    done = p
                    curve[p] = metrics(model)
                curves.append(curve)
        line = []
        for p in extra:
            o = np.array([c[p] for c in curves])
            s = f"{p}: {o[:, 0].mean():.4f}/{o[:, 1].mean():.4f}"
            if len(o) > 3:
                s += f" se {o[:, 0].std(ddof=1) / math.sqrt(len(o)):.4f}"
            if p in ref:
                s += f" ({o[:, 0].mean() - ref[p][:, 0].mean():+.4f}/{o[:, 1].mean() - ref[p][:, 1].mean():+.4f})"
            line.append(s)
        print(f"{name:18s} " + "  ".join(line), flush=True)


if __name__ == "__main__":
    main()
