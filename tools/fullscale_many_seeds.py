"""The snapshot schedules of STREAM against STRICT at the FULL ML-20M shape with more seeds than
tests/test_gpu_fullscale_parity.py runs (its set, its run functions):
  python tools/fullscale_many_seeds.py N_SEEDS [FIRST_SEED [sgd|adam]]
STRICT takes ~2 s per epoch here, STREAM 16 ms.  Output of r03: profiles/r03_fullscale_many_seeds.txt"""
import math
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "revisit-bpr_amd"))
sys.path.insert(0, str(ROOT))
import tests.test_gpu_fullscale_parity as T  # noqa: E402
from revisit_bpr.datasets import synthetic  # noqa: E402

n, first = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1
data = synthetic.generate_latent(136677, 20108, 9_700_000, factors=16, strength=1.2, median_per_user=37,
                                 min_per_user=5, seed=13, eval_users=10_000, item_skew=1.2, item_shift=60.0)
dev = torch.device("cuda")
t = {k: torch.from_numpy(getattr(data, k)).to(dev)
     for k in ("users", "items", "indptr", "indices", "eval_users", "eval_indptr", "eval_items")}
seeds = range(first, first + n)
mode = sys.argv[3] if len(sys.argv) > 3 else "sgd"
if mode == "adam":  # BASELINE configs[4]'s optimizer through the batched stream (k_vstream)
    EPOCHS = 4
    make_opt = lambda p: torch.optim.Adam(p, lr=0.002, betas=(0.1, 0.999))  # noqa: E731
    desc = "Adam(0.1, 0.999) lr 0.002"
    paths = (("BATCHED-adam", "batched", {}),)
else:
    EPOCHS = 6
    make_opt = lambda p: torch.optim.SGD(p, lr=0.05)  # noqa: E731
    desc = "SGD lr 0.05"
    paths = (("STREAM[reference schedule]", "stream", {}),
             ("STREAM[lag 1, sort on 64 masked CUs]", "stream", {"refresh_lag": 1.0, "refresh_cus": 64}),
             ("BATCHED-sgd", "batched", {}))
strict = np.stack([T.run(data, t, "strict", make_opt, EPOCHS, s) for s in seeds])
print("strict done", flush=True)
for name, kind, sched in paths:
    ours = np.stack([T.run(data, t, kind, make_opt, EPOCHS, s, **sched) for s in seeds])
    for k, key in ((0, "ndcg@100"), (1, "recall@20")):
        for epoch in (min(2, EPOCHS - 2), EPOCHS - 1):
            o, r = ours[:, epoch, k], strict[:, epoch, k]
            se = math.sqrt(o.var(ddof=1) / len(o) + r.var(ddof=1) / len(r))
            print(f"{name} vs STRICT, ML-20M shape d=128 {desc}, {key} after epoch {epoch + 1}: "
                  f"{o.mean():.4f} vs {r.mean():.4f} (n={len(o)}, sd {o.std(ddof=1):.4f} / {r.std(ddof=1):.4f}) "
                  f"diff {o.mean() - r.mean():+.4f} se {se:.4f}", flush=True)
