#!/usr/bin/env python
"""r6: what does one evaluation of the reference's ML-20M protocol cost (10,000 held-out users x 20,108 items, the 14
metrics of configs/RQ2/neg-sampling/ada-sampling-ml-20m.yaml.j2) through `evaluate_topk`, piece by piece?"""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "revisit-bpr_amd")]
from revisit_bpr.datasets import synthetic
from revisit_bpr.evaluation import evaluate_topk

dev = torch.device("cuda")
data = synthetic.generate_named("ml-20m", eval_users=10_000, seed=3)
U, I, d = data.num_users, data.num_items, 128
g = torch.Generator().manual_seed(1)
P = ((torch.rand(U, d, generator=g) - 0.5) / d).to(dev)
Q = ((torch.rand(I, d, generator=g) - 0.5) / d).to(dev)
t = {k: torch.from_numpy(getattr(data, k)).to(dev) for k in ("eval_users", "eval_indptr", "eval_items", "indptr", "indices")}


def timed(label, fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    print(f"{label:50s} {(time.perf_counter() - t0) / n * 1e3:8.2f} ms", flush=True)
    return out


ks = (5, 10, 20, 50, 100)
for blk in (2048, 4096, 10000):
    timed(f"evaluate_topk ks={ks} block {blk}", lambda: evaluate_topk(P, Q, None, t["eval_users"], t["eval_indptr"], t["eval_items"], t["indptr"], t["indices"], ks=ks, block=blk))
    a = timed(f"evaluate_topk + auc      block {blk}", lambda: evaluate_topk(P, Q, None, t["eval_users"], t["eval_indptr"], t["eval_items"], t["indptr"], t["indices"], ks=ks, block=blk, auc=True))
users = t["eval_users"][:4096].long()
timed("  GEMM [4096,128]x[128,20108]", lambda: P[users] @ Q.T)
lg = P[users] @ Q.T
timed("  topk 100 of [4096, 20108]", lambda: torch.topk(lg, 100, dim=1))
timed("  sort of [4096, 20108]", lambda: torch.sort(lg, dim=-1))
timed("  zeros [4096, 20108] fp32", lambda: torch.zeros(4096, I, device=dev))
print("auc", a.get("auc"), "ndcg@100", a["ndcg@100"])
