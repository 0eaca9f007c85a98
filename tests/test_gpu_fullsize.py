"""Full-size runs (BASELINE.json configs[2], ML-20M shape, d=128) checked through size-independent
properties — the oracle would take hours at this size:
  * bpr_plan_epoch is a permutation of the 9.55 M training pairs, chunks are grouped by user;
  * one STREAM epoch per sampler: every triple is processed exactly once (count), every sampled
    negative is a valid draw (never item 0, never a seen item), the mean loss falls below ln 2,
    tables stay finite, pad rows stay zero;
  * the refresh snapshot is a per-factor permutation sorted by the item table's column.
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ml20m():
    from revisit_bpr.datasets import synthetic

    return synthetic.generate_named("ml-20m", eval_users=10_000, seed=13)


def _engine(data, d=128, lr=0.05):
    from revisit_bpr import engine as eng

    g = torch.Generator().manual_seed(13)
    P = (torch.rand(data.num_users, d, generator=g) - 0.5) / d
    Q = (torch.rand(data.num_items, d, generator=g) - 0.5) / d
    P[0] = 0
    Q[0] = 0
    e = eng.Engine(P.cuda(), Q.cuda())
    e.set_reg(0.0016, 0.0001, 0.00375)
    e.set_optimizer(eng.OPT_SGD, lr=lr)
    e.bind_seen_csr(torch.from_numpy(data.indptr).cuda(), torch.from_numpy(data.indices).cuda())
    return e


def test_plan_epoch_full_size(ml20m):
    e = _engine(ml20m)
    u, i = torch.from_numpy(ml20m.users).cuda(), torch.from_numpy(ml20m.items).cuda()
    chunk = int(ml20m.num_items * math.log(ml20m.num_items) / 256) * 256
    pu, pi = e.plan_epoch(u, i, chunk, seed=1)
    key_in = u.long() * ml20m.num_items + i.long()
    key_out = pu.long() * ml20m.num_items + pi.long()
    assert torch.equal(torch.sort(key_in).values, torch.sort(key_out).values)
    n = u.numel()
    starts = torch.arange(0, n, chunk, device="cuda")
    diffs = pu[1:].long() - pu[:-1].long()
    boundary = torch.zeros(n - 1, dtype=torch.bool, device="cuda")
    boundary[starts[1:] - 1] = True
    assert bool((diffs[~boundary] >= 0).all())  # sorted by user inside every chunk


@pytest.mark.parametrize("sampler", ["uniform", "adaptive"])
def test_stream_epoch_full_size(ml20m, sampler):
    from revisit_bpr import engine as eng

    data = ml20m
    e = _engine(data)
    e.set_stream_opts(True, 8)
    u, i = torch.from_numpy(data.users).cuda(), torch.from_numpy(data.items).cuda()
    I = data.num_items
    chunk = int(I * math.log(I) / 256) * 256
    pu, pi = e.plan_epoch(u, i, chunk, seed=3)
    neg = torch.zeros_like(pu)
    sc = torch.zeros(4, device="cuda")
    kind = eng.NEG_ADAPTIVE if sampler == "adaptive" else eng.NEG_UNIFORM
    for lo in range(0, pu.numel(), chunk):
        hi = min(lo + chunk, pu.numel())
        if kind == eng.NEG_ADAPTIVE:
            e.adaptive_refresh()
        e.train_stream(pu[lo:hi], pi[lo:hi], sampler=kind, neg=neg[lo:hi], adaptive_p=0.01, seed=5,
                       offset=lo, scalars=sc)
    torch.cuda.synchronize()
    assert int(sc[3]) == data.nnz
    assert float(sc[0] / sc[3]) < math.log(2.0)  # learning: mean -log sigma(x) below the x=0 value
    assert torch.isfinite(e.P).all() and torch.isfinite(e.Q).all()
    assert not e.P[0].any() and not e.Q[0].any()
    assert int(neg.min()) >= 1 and int(neg.max()) < I
    # membership of every (user, negative) pair in the seen CSR, via sorted keys
    seen_keys = torch.from_numpy(
        np.repeat(np.arange(data.num_users, dtype=np.int64), np.diff(data.indptr)) * I
        + data.indices.astype(np.int64)).cuda()
    q = pu.long() * I + neg.long()
    pos = torch.searchsorted(seen_keys, q).clamp(max=seen_keys.numel() - 1)
    assert not bool((seen_keys[pos] == q).any())
    if sampler == "adaptive":  # a fresh snapshot is a per-factor permutation sorted by the column
        e.adaptive_refresh()
        order, sigma = e.adaptive_snapshot()
        assert bool((torch.sort(order, dim=1).values == torch.arange(I, device="cuda", dtype=torch.int32)).all())
        col = e.Q[:, 7][order[7].long()]
        assert bool((col[1:] <= col[:-1]).all())
        assert float(sigma.min()) > 0
