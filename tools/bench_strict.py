#!/usr/bin/env python
"""Throughput of the STRICT epoch driver (exact reference mini-batch semantics, any optimizer):
    python tools/bench_strict.py --workload yelp --dim 128 --opt adam
Not the headline bench (bench.py); used for BASELINE config 5 (Adam path)."""
import argparse
import math
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "revisit-bpr_amd")]
import torch  # noqa: E402

from revisit_bpr import engine as eng  # noqa: E402
from revisit_bpr.datasets import synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="yelp")
ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--opt", default="adam")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--sampler", default="adaptive")
ap.add_argument("--triples", type=int, default=1_000_000)
ap.add_argument("--beta1", type=float, default=0.1)
ap.add_argument("--start-step", type=int, default=0, help="pretend this many optimizer steps were taken")
a = ap.parse_args()
data = synthetic.generate_named(a.workload, seed=13)
dev = torch.device("cuda")
g = torch.Generator().manual_seed(13)
P = ((torch.rand(data.num_users, a.dim, generator=g) - 0.5) / a.dim).to(dev)
Q = ((torch.rand(data.num_items, a.dim, generator=g) - 0.5) / a.dim).to(dev)
e = eng.Engine(P, Q)
e.set_reg(0.0025, 0.0025, 0.00025)
kind = {"sgd": eng.OPT_SGD, "adam": eng.OPT_ADAM, "rmsprop": eng.OPT_RMSPROP, "momentum": eng.OPT_MOMENTUM}[a.opt]
e.set_optimizer(kind, lr=0.001, betas=(a.beta1, 0.999), momentum=0.9, nesterov=True, alpha=0.9)
e.alloc_opt_state()
if a.start_step:
    e.set_step(a.start_step)
e.bind_seen_csr(torch.from_numpy(data.indptr).to(dev), torch.from_numpy(data.indices).to(dev))
e.adaptive_refresh()
n = min(a.triples, data.nnz)
perm = torch.randperm(data.nnz, device=dev)[:n]
users = torch.from_numpy(data.users).to(dev)[perm].contiguous()
items = torch.from_numpy(data.items).to(dev)[perm].contiguous()
every = int(data.num_items * math.log(data.num_items) / a.batch)
smp = eng.NEG_ADAPTIVE if a.sampler == "adaptive" else eng.NEG_UNIFORM
sc = torch.zeros(4, device=dev)
e.train_strict(users[:a.batch * 20], items[:a.batch * 20], a.batch, sampler=smp, seed=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
e.train_strict(users, items, a.batch, sampler=smp, adaptive_p=0.01, seed=1, refresh_every=every, scalars=sc)
e.flush_lazy()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{a.workload} d={a.dim} {a.opt} {a.sampler} B={a.batch}: {n} triples in {dt * 1e3:.1f} ms = "
      f"{n / dt / 1e6:.2f} M triples/s ({dt / (n / a.batch) * 1e6:.1f} us/step), mean loss {float(sc[0] / sc[3]):.4f}")
