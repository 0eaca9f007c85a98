"""The LDS tier of k_stream's hot block (r6: `bpr_set_hot_lds`, csrc/bpr_hotlds.hip) against the CPU oracle and
against the plain kernel.

What the tier may change is WHEN a workgroup sees the other workgroups' updates of the hottest rows (one launch
late); what it may not change:
  * with one group in flight it is exactly sequential SGD (the oracle's `train_stream_seq`), sampler included;
  * nothing is lost or counted twice: the launch's deltas are the exact sums (first-order test on a zero table);
  * sampled negatives are valid at full concurrency (never the pad item, never a seen one; uniform picks are
    the plain kernel's);
  * the fused cut after the launch sees the table with every workgroup's flush folded in;
  * item_bias, loss statistics and the pad rows behave as in the plain kernel.
"""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from test_gpu_parity import close, dev, make_engine, maxerr  # noqa: E402


def skewed_problem(U, I, d, n, seed, max_seen=40):
    rng = np.random.default_rng(seed)
    P = rng.normal(0, 0.1, (U, d)).astype(np.float32)
    Q = rng.normal(0, 0.1, (I, d)).astype(np.float32)
    P[0] = 0
    Q[0] = 0
    lens = rng.integers(1, max_seen, U)
    lens[0] = 0
    rows = [np.sort(rng.choice(np.arange(1, I), size=int(k), replace=False)).astype(np.int32) for k in lens]
    indptr = np.zeros(U + 1, np.int64)
    indptr[1:] = np.cumsum(lens)
    users = rng.integers(1, U, n).astype(np.int32)
    pos = (1 + (rng.zipf(1.3, n) % (I - 1))).astype(np.int32)  # a few rows take most of the positives
    return P, Q, indptr, np.concatenate(rows), users, pos, rng


@pytest.mark.parametrize("seen", ["", "list"])
@pytest.mark.parametrize("d,run_len,sampler,bias", [(256, 8, 1, False), (256, 3, 2, False), (512, 5, 0, True),
                                                     (1024, 8, 2, True)])
def test_lds_tier_sequential_equals_b1_sgd(d, run_len, sampler, bias, seen, monkeypatch):
    """(seen = "list": the groups' staged sorted seen lists beside the rows instead of the I-bit bitmaps — what the
    tier takes by itself for item tables too large for bitmaps.)
    One group in flight (G = 64: a whole wave) == the oracle's sequential SGD in planned order, three launches
    in a row (flush -> fold -> next launch reads the folded table), with 5 of the 12 hot rows in LDS and the other
    7 in the global block."""
    if seen:
        if sampler == 0:
            pytest.skip("given negatives: no seen structure")
        monkeypatch.setenv("BPR_SEEN", seen)
    P, Q, indptr, indices, users, pos, rng = skewed_problem(120, 80, d, 1400, d + run_len)
    b = (rng.normal(0, 0.1, Q.shape[0]).astype(np.float32)) if bias else None
    reg = (0.01, 0.02, 0.03)
    e = make_engine(P, Q, b, reg)
    e.bind_seen_csr(dev(indptr), dev(indices))
    e.set_optimizer(kind=0, lr=0.05)
    e.set_stream_opts(True, run_len)
    e.set_hot_rows(12, 1)
    e.set_hot_lds(8, always=True)  # (asks for 8 ...)
    pu, pp = e.plan_epoch(dev(users), dev(pos), chunk=len(users), seed=3)
    Po, Qo, bo = P.copy(), Q.copy(), None if b is None else b.copy()
    upl, ppl = pu.cpu().numpy(), pp.cpu().numpy()
    given = rng.integers(1, Q.shape[0], len(users)).astype(np.int32)
    sco = np.zeros(4)
    sc = torch.zeros(4, device="cuda")
    for launch in range(3):
        if sampler == 2:
            e.adaptive_refresh()
            QT, sigma = oracle.adaptive_stats(Qo)
            snap = dict(order=oracle.adaptive_order(QT), sigma=sigma, adaptive_p=0.05)
        else:
            snap = {}
        negs = dev(given) if sampler == 0 else torch.zeros_like(pu)
        e.train_stream(pu, pp, sampler=sampler, neg=negs, adaptive_p=0.05, seed=11, offset=launch * len(users),
                       max_inflight=1, scalars=sc)
        assert e.stream_lds_rows() == 8
        neg_o = given.copy() if sampler == 0 else np.zeros(len(users), np.int32)
        sco += oracle.train_stream_seq(Po, Qo, bo, upl, ppl, neg_o, sampler, 0.05, reg, indptr=indptr,
                                       indices=indices, seed=11, offset=launch * len(users), **snap)
        if sampler != 0:
            assert np.array_equal(negs.cpu().numpy(), neg_o), launch
        assert close(e.Q.cpu().numpy(), Qo, 1e-5), (launch, maxerr(e.Q.cpu().numpy(), Qo))
    assert close(e.P.cpu().numpy(), Po, 1e-5), maxerr(e.P.cpu().numpy(), Po)
    if bias:
        assert close(e.item_bias.cpu().numpy(), bo, 1e-5)
    assert close(sc.cpu().numpy()[:3], sco[:3], 1e-4) and int(sc[3]) == 3 * len(users)


@pytest.mark.parametrize("hot_rows,replicas,lds", [(16, 1, 8), (50, 4, 50), (50, 1, 20)])
def test_lds_tier_loses_nothing_under_chip_wide_contention(hot_rows, replicas, lds):
    """100k triples on 50 item rows from every CU at once, part of the rows in LDS: on a zero item table with a
    tiny learning rate every triple contributes lr x (gradient at the initial point) to first order, so the
    table after the launch is the exact sum — whatever a workgroup saw of the others meanwhile."""
    U, I, d, n = 20000, 60, 128, 100_000
    rng = np.random.default_rng(0)
    P = ((rng.random((U, d)) - 0.5) * 0.2).astype(np.float32)
    P[0] = 0
    Q0 = np.zeros((I, d), np.float32)
    users = rng.integers(1, U, size=n).astype(np.int32)
    pos = rng.integers(1, 51, size=n).astype(np.int32)
    neg = rng.integers(1, 51, size=n).astype(np.int32)
    lr = 1e-4
    e = make_engine(P, Q0, None, (0.0, 0.0, 0.0))
    e.set_optimizer(kind=0, lr=lr)
    e.set_hot_rows(hot_rows, replicas)
    e.set_hot_lds(lds, always=True)
    e.set_stream_opts(True, 8)
    pu, pi = e.plan_epoch(dev(users), dev(pos), n, seed=3)
    back = {(int(a), int(b)): k for k, (a, b) in enumerate(zip(users, pos))}
    ng = np.asarray([neg[back[(int(a), int(b))]] for a, b in zip(pu.cpu().numpy(), pi.cpu().numpy())], np.int32)
    sc = torch.zeros(4, device="cuda")
    e.train_stream(pu, pi, sampler=0, neg=dev(ng), scalars=sc)
    assert e.stream_lds_rows() == min(lds, hot_rows)
    _, g, _ = oracle.dense_grad(P, Q0, None, pu.cpu().numpy(), pi.cpu().numpy(), ng, (0.0, 0.0, 0.0))
    dQ = e.Q.cpu().numpy().astype(np.float64) / -lr
    s0 = np.abs(g).max()
    assert s0 > 3 and np.abs(dQ - g).max() < 0.005 * s0, np.abs(dQ - g).max() / s0
    assert int(sc[3]) == n


@pytest.mark.parametrize("d,sampler,n,cut", [(128, 2, 120_000, True), (128, 1, 60_000, True), (64, 2, 90_000, True),
                                              (256, 2, 40_000, True), (32, 0, 50_000, True), (128, 2, 120_000, "async"),
                                              (64, 1, 70_000, "async")])
def test_lds_tier_cut_at_full_concurrency_sees_the_final_table(d, sampler, n, cut):
    """(cut = "async": the transpose on the side stream, the fold on the launch stream — with nothing running beside
    it here, the same exact snapshot.)
    The fused cut behind an LDS-tier launch: the snapshot committed afterwards is the oracle's order of the
    item table as the launch left it — every workgroup's flush folded in — launch after launch; the statistics
    count every triple; sampled negatives are valid (uniform: the plain kernel's picks triple by triple; adaptive
    picks read the live user rows, which at lr 0.05 and 20 triples per user have moved apart within the launch)."""
    P, Q, indptr, indices, users, pos, rng = skewed_problem(6000, 3000, d, n, d + n)
    engines = []
    for lds in (0, 64):
        e = make_engine(P, Q, None, (0.01, 0.02, 0.03))
        e.bind_seen_csr(dev(indptr), dev(indices))
        e.set_optimizer(kind=0, lr=0.05)
        e.set_stream_opts(True, 0)
        e.set_hot_lds(lds, always=True)
        engines.append(e)
    e0, e = engines
    pu, pi = e.plan_epoch(dev(users), dev(pos), n, seed=3)
    pu0, pi0 = e0.plan_epoch(dev(users), dev(pos), n, seed=3)
    assert torch.equal(pu, pu0) and torch.equal(pi, pi0)
    given = dev(rng.integers(1, Q.shape[0], n).astype(np.int32))
    sc = torch.zeros(4, device="cuda")
    e.adaptive_refresh()
    e0.adaptive_refresh()
    for launch in range(3):
        neg = given if sampler == 0 else torch.zeros_like(pu)
        e.train_stream(pu, pi, sampler=sampler, neg=neg, adaptive_p=0.05, seed=5, offset=launch * n, scalars=sc,
                       cut=cut)
        assert e.stream_lds_rows() > 0
        if launch == 0 and sampler != 0:
            neg0 = torch.zeros_like(pu)
            e0.train_stream(pu0, pi0, sampler=sampler, neg=neg0, adaptive_p=0.05, seed=5, offset=0)
            assert e0.stream_lds_rows() == 0
            if sampler == 1:  # uniform picks do not depend on the model: the same, triple by triple
                assert torch.equal(neg, neg0)
            un, nn = pu.cpu().numpy(), neg.cpu().numpy()
            for t in range(0, n, 211):
                assert nn[t] != 0 and nn[t] not in indices[indptr[un[t]]:indptr[un[t] + 1]]
        Qnow = e.Q.cpu().numpy()
        e.adaptive_refresh_begin()
        e.adaptive_refresh_commit()
        QT, sig = oracle.adaptive_stats(Qnow)
        got_o, got_s = e.adaptive_snapshot()
        assert np.array_equal(got_o.cpu().numpy(), oracle.adaptive_order(QT)), launch
        assert close(got_s.cpu().numpy(), sig, 1e-5)
        assert int(sc[3]) == (launch + 1) * n
    assert torch.isfinite(e.Q).all() and torch.isfinite(e.P).all() and torch.isfinite(sc).all()
    assert not e.P[0].any() and not e.Q[0].any()


def test_lds_tier_follows_the_plain_kernel_at_a_small_learning_rate():
    """Same triples, same given negatives, lr small enough that one launch's staleness is second order (the top row
    takes a third of the 150 k positives: lr x updates per row is what sets it): the two kernels leave the same
    tables to 1 % of the largest update; and the tier is NOT taken where it cannot be (no hot block, a launch that
    does not fill the chip without `always`)."""
    d, n = 128, 150_000
    P, Q, indptr, indices, users, pos, rng = skewed_problem(8000, 2000, d, n, 7)
    neg = rng.integers(1, Q.shape[0], n).astype(np.int32)
    out = []
    for lds in (0, 128):
        e = make_engine(P, Q, None, (0.01, 0.02, 0.03))
        e.set_optimizer(kind=0, lr=2e-6)
        e.set_stream_opts(True, 8)
        e.set_hot_lds(lds, always=True)
        pu, pi = e.plan_epoch(dev(users), dev(pos), n, seed=3)
        back = {(int(a), int(b)): k for k, (a, b) in enumerate(zip(users, pos))}
        ng = np.asarray([neg[back[(int(a), int(b))]] for a, b in zip(pu.cpu().numpy(), pi.cpu().numpy())], np.int32)
        e.train_stream(pu, pi, sampler=0, neg=dev(ng))
        assert (e.stream_lds_rows() > 0) == (lds > 0)
        out.append((e.P.cpu().numpy(), e.Q.cpu().numpy()))
    moved = np.abs(out[0][1] - Q).max()
    assert moved > 2e-4
    assert np.abs(out[0][1] - out[1][1]).max() < 1e-2 * moved, np.abs(out[0][1] - out[1][1]).max() / moved
    assert np.abs(out[0][0] - out[1][0]).max() < 1e-2 * moved
    # not taken: no hot block
    e = make_engine(P, Q, None, (0.01, 0.02, 0.03))
    e.set_optimizer(kind=0, lr=1e-4)
    e.set_hot_lds(128, always=True)
    e.train_stream(dev(users), dev(pos), sampler=0, neg=dev(neg))
    assert e.stream_lds_rows() == 0
    # not taken: a launch that does not fill the chip twice, unless forced
    e.set_hot_lds(128, always=False)
    e.set_stream_opts(True, 8)
    pu, pi = e.plan_epoch(dev(users[:20_000]), dev(pos[:20_000]), 20_000, seed=3)
    e.train_stream(pu, pi, sampler=0, neg=dev(neg[:20_000]))
    assert e.stream_lds_rows() == 0
