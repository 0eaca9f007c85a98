#!/usr/bin/env python
"""STREAM with deferred positives (bpr_set_defer_positives 0 | 1 | 2) against STRICT — the
reference's mini-batch loop — at the full ML-20M shape (the set-up of
tests/test_gpu_fullscale_parity.py): nDCG@100 / Recall@20 per epoch, seed means.
    python tools/defer_study.py [lr] [n_seeds] [epochs]"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "revisit-bpr_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
import test_gpu_fullscale_parity as T  # noqa: E402

lr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
modes = sys.argv[4].split(",") if len(sys.argv) > 4 else ["strict", "0", "1", "2"]
eval_every = int(sys.argv[5]) if len(sys.argv) > 5 else 1


def run_stream(seed):
    """T.run for STREAM, evaluating every `eval_every` epochs only"""
    from revisit_bpr.fast import StreamTrainer

    model = T.fresh_model(data)
    tr = StreamTrainer(model, t["users"], t["items"], t["indptr"], t["indices"], lr=lr, sampler="adaptive",
                       adaptive_p=T.P_GEO, batch_size=T.B, seed=seed)
    curve = []
    for ep in range(epochs):
        tr.train_epoch()
        if (ep + 1) % eval_every == 0:
            m = T.metrics(model, t)
            curve.append((m["ndcg@100"], m["recall@20"]))
    return np.array(curve)


from revisit_bpr.datasets import synthetic  # noqa: E402

data = synthetic.generate_latent(136677, 20108, 9_700_000, factors=16, strength=1.2, median_per_user=37,
                                 min_per_user=5, seed=13, eval_users=10_000, item_skew=1.2, item_shift=60.0)
dev = torch.device("cuda")
t = {k: torch.from_numpy(getattr(data, k)).to(dev)
     for k in ("users", "items", "indptr", "indices", "eval_users", "eval_indptr", "eval_items")}
make_opt = lambda p: torch.optim.SGD(p, lr=lr)  # noqa: E731
print(f"lr={lr} seeds={n_seeds} epochs={epochs}", flush=True)
res = {}
for mode in modes:
    if mode == "strict":
        os.environ.pop("BPR_DEFER_POS", None)
        curves = np.stack([T.run(data, t, "strict", make_opt, epochs, s) for s in range(1, n_seeds + 1)])
    else:
        os.environ["BPR_DEFER_POS"] = mode
        curves = np.stack([run_stream(s) for s in range(1, n_seeds + 1)])
    res[mode] = curves
    name = "STRICT" if mode == "strict" else f"STREAM defer={mode}"
    print(f"{name:16s} nDCG@100 " + " ".join(f"{v:.4f}" for v in curves[:, :, 0].mean(0)) +
          "   (+-" + f"{curves[:, -1, 0].std(ddof=1) if n_seeds > 1 else 0:.4f})", flush=True)
    print(f"{'':16s} Rec@20   " + " ".join(f"{v:.4f}" for v in curves[:, :, 1].mean(0)), flush=True)
if "strict" in res:
    for mode in modes:
        if mode != "strict":
            dlt = res[mode].mean(0) - res["strict"].mean(0)
            print(f"defer={mode} - STRICT: nDCG " + " ".join(f"{v:+.4f}" for v in dlt[:, 0]) +
                  " | Rec " + " ".join(f"{v:+.4f}" for v in dlt[:, 1]))
