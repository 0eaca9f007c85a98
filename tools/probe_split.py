"""Timing probe: standalone sampling kernels vs the fused stream kernel (ML-20M shape, d=128)."""
import math, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "revisit-bpr_amd")]
import torch
from revisit_bpr import engine as eng
from revisit_bpr.datasets import synthetic

data = synthetic.generate_named("ml-20m", eval_users=10000, seed=13)
dev = torch.device("cuda"); d = 128
g = torch.Generator().manual_seed(13)
P = ((torch.rand(data.num_users, d, generator=g) - 0.5) / d).to(dev)
Q = ((torch.rand(data.num_items, d, generator=g) - 0.5) / d).to(dev)
e = eng.Engine(P, Q); e.set_reg(0.0016, 0.0001, 0.00375); e.set_optimizer(eng.OPT_SGD, lr=0.001)
e.bind_seen_csr(torch.from_numpy(data.indptr).to(dev), torch.from_numpy(data.indices).to(dev))
I = data.num_items; chunk = int(I * math.log(I) / 256) * 256
su, si = torch.from_numpy(data.users).to(dev), torch.from_numpy(data.items).to(dev)
e.set_stream_opts(True, 8)
pu, pi = e.plan_epoch(su, si, chunk, 1)
e.adaptive_refresh()
u, i = pu[:chunk].contiguous(), pi[:chunk].contiguous()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
neg = e.sample_adaptive(u, 0.01, seed=1)
print("k_sample adaptive (CSR search, 1 triple/group): %.3f ms" % timeit(lambda: e.sample_adaptive(u, 0.01, seed=1)))
print("k_sample uniform                              : %.3f ms" % timeit(lambda: e.sample_uniform(u, seed=1)))
print("k_stream GIVEN                                : %.3f ms" % timeit(lambda: e.train_stream(u, i, sampler=0, neg=neg)))
print("k_stream ADAPTIVE                             : %.3f ms" % timeit(lambda: e.train_stream(u, i, sampler=2, adaptive_p=0.01, seed=1)))
print("k_stream UNIFORM                              : %.3f ms" % timeit(lambda: e.train_stream(u, i, sampler=1, seed=1)))
print("refresh                                       : %.3f ms" % timeit(lambda: e.adaptive_refresh()))
