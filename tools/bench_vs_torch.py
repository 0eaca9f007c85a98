#!/usr/bin/env python
"""BASELINE config 2: the fused HIP path against the same model as dense PyTorch-ROCm autograd +
torch.optim on the same GPU (`set_backend("torch")`, a restatement of the reference's forward:
revisit_bpr/models/bpr/model.py:48-93), Netflix shape, d=64, uniform negatives, SGD.

    python tools/bench_vs_torch.py [--workload netflix --dim 64 --batches 400]

Three ways through the same epoch slice:
  torch   model(batch) / backward / torch.optim.SGD.step   (dense [U,d] + [I,d] gradients)
  strict  the same loop, fused engine behind the same API  (reference-exact mini-batches)
  stream  StreamTrainer                                     (the throughput path)
"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "revisit-bpr_amd")]
import torch  # noqa: E402

from revisit_bpr.datasets import synthetic  # noqa: E402
from revisit_bpr.fast import StreamTrainer  # noqa: E402
from revisit_bpr.models import BPR  # noqa: E402
from revisit_bpr.models.bpr import MF, set_backend  # noqa: E402
from revisit_bpr.modules import UniformSampler  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="netflix")
ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--batches", type=int, default=400)
a = ap.parse_args()
dev = torch.device("cuda")
data = synthetic.generate_named(a.workload, seed=13)
t = {k: torch.from_numpy(getattr(data, k)).to(dev) for k in ("users", "items", "indptr", "indices")}
reg = {"user": 0.0016, "item": 0.0001, "neg": 0.00375}
n = min(a.batches * a.batch, data.nnz)
perm = torch.randperm(data.nnz, device=dev, generator=torch.Generator(device=dev).manual_seed(1))[:n]
users, items = t["users"][perm].long(), t["items"][perm].long()
neg = torch.randint(1, data.num_items, (n, 1), device=dev)


def model():
    torch.manual_seed(13)
    return BPR(fuse_forward=True, reg_alphas=reg,
               logits_model=MF(torch.nn.Embedding(data.num_users, a.dim, padding_idx=0),
                               torch.nn.Embedding(data.num_items, a.dim, padding_idx=0))).to(dev)


def api_loop(backend):
    set_backend(backend)
    try:
        m = model()
        opt = torch.optim.SGD(m.parameters(), lr=0.05)
        m.train()
        for rep in range(2):  # first pass warms up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for lo in range(0, n, a.batch):
                out = m({"user": users[lo:lo + a.batch], "item": items[lo:lo + a.batch].unsqueeze(1),
                         "neg": neg[lo:lo + a.batch]})
                out["loss"].backward()
                opt.step()
                opt.zero_grad()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        return n / dt
    finally:
        set_backend("hip")


r_torch = api_loop("torch")
r_strict = api_loop("hip")
tr = StreamTrainer(model(), t["users"], t["items"], t["indptr"], t["indices"], lr=0.05,
                   sampler="uniform", batch_size=a.batch)
tr.train_epoch()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    tr.train_epoch()
torch.cuda.synchronize()
r_stream = 5 * data.nnz / (time.perf_counter() - t0)
print(f"{a.workload} d={a.dim} B={a.batch} uniform SGD, triples/s:  torch-ROCm dense {r_torch:,.0f}   "
      f"fused STRICT behind the same API {r_strict:,.0f} ({r_strict / r_torch:.1f}x)   "
      f"STREAM {r_stream:,.0f} ({r_stream / r_torch:.0f}x)")
