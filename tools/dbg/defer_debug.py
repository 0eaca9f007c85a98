import sys
import numpy as np
import torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/revisit-bpr_amd"); sys.path.insert(0, "/root/repo/tests")
import oracle
from revisit_bpr.datasets import synthetic
from test_gpu_parity import make_engine, dev

d, run_len, mode = 128, 8, 2
data = synthetic.generate(150, 90, 1500, median_per_user=8, seed=d + run_len)
rng = np.random.default_rng(d)
P = ((rng.random((data.num_users, d)) - 0.5) * 0.5).astype(np.float32)
Q = ((rng.random((data.num_items, d)) - 0.5) * 0.5).astype(np.float32)
P[0] = 0; Q[0] = 0
reg = (0.01, 0.02, 0.03)
for hot in (0, 8):
    e = make_engine(P, Q, None, reg)
    e.bind_seen_csr(dev(data.indptr), dev(data.indices))
    e.set_optimizer(kind=0, lr=0.05)
    e.set_stream_opts(True, run_len)
    e.set_hot_rows(hot, 1)
    e.set_defer_positives(mode)
    pu, pp = e.plan_epoch(dev(data.users), dev(data.items), chunk=data.nnz, seed=3)
    given = rng.integers(1, data.num_items, data.nnz).astype(np.int32)
    negs = dev(given)
    e.train_stream(pu, pp, sampler=0, neg=negs, seed=11, max_inflight=1)
    Po, Qo = P.copy(), Q.copy()
    oracle.train_stream_seq_deferred(Po, Qo, None, pu.cpu().numpy(), pp.cpu().numpy(), given.copy(), 0, 0.05, reg,
                                     indptr=data.indptr, indices=data.indices, seed=11)
    Qg = e.Q.cpu().numpy()
    err = np.abs(Qg - Qo).max(axis=1)
    cnt = np.bincount(pp.cpu().numpy(), minlength=data.num_items)
    ncnt = np.bincount(given, minlength=data.num_items)
    print("hot", hot, "maxerr P", np.abs(e.P.cpu().numpy() - Po).max(), "Q", err.max())
    bad = np.argsort(-err)[:8]
    for i in bad:
        print("  item", i, "err", err[i], "pos cnt", cnt[i], "neg cnt", ncnt[i], "move", np.abs(Qo[i] - Q[i]).max())
