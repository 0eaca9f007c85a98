"""Distribution of the DEVICE samplers' output against the reference's FORMULAE (not against the
oracle's draws — those share the Philox scheme with the kernels):

  * uniform (neg_samplers.py:31-37, 135-141): uniform over the unseen items, never item 0;
  * weighted static sampling (experiments/bpr/exp.py:85-91, 282-293): w_i / sum of w over unseen;
  * adaptive (neg_samplers.py:74-124): factor f ~ |p_uf| sigma_f; r ~ Geometric(p) on {1, 2, ...}
    clamped to the number of unseen items; rank r-1 from the top of factor f's snapshot order when
    p_uf > 0, else from the bottom; sigma_f = unbiased std of Q[1:, f].

Chi-square goodness of fit with 5-sigma acceptance (statistical tests: they run after the
deterministic suites, tests/conftest.py)."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from test_gpu_parity import dev, make_engine  # noqa: E402


def chi2_ok(obs, expect, n):
    keep = expect > 0
    assert obs[~keep].sum() == 0
    chi2 = (((obs - n * expect) ** 2)[keep] / (n * expect[keep])).sum()
    dof = keep.sum() - 1
    return chi2 < dof + 5 * np.sqrt(2 * dof), (chi2, dof)


def small_problem(U=40, I=90, d=32, seed=0):
    rng = np.random.default_rng(seed)
    P = rng.standard_normal((U, d)).astype(np.float32) * 0.3
    Q = rng.standard_normal((I, d)).astype(np.float32) * 0.3
    P[0] = 0
    Q[0] = 0
    rows = [np.sort(rng.choice(np.arange(1, I), size=int(rng.integers(0, 30)), replace=False))
            for _ in range(U)]
    rows[0] = np.zeros(0, np.int64)
    indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    indices = np.concatenate(rows).astype(np.int32)
    return P, Q, rows, indptr, indices


@pytest.mark.parametrize("weighted", [False, True])
def test_uniform_and_weighted_draws_follow_the_reference_weights(weighted):
    P, Q, rows, indptr, indices = small_problem()
    U, I = P.shape[0], Q.shape[0]
    e = make_engine(P, Q)
    e.bind_seen_csr(dev(indptr), dev(indices))
    rng = np.random.default_rng(1)
    w = np.ones(I)
    if weighted:
        w = rng.integers(1, 300, I).astype(np.float64) ** 0.6  # count ** neg_sampling_alpha
        e.bind_item_weights(torch.from_numpy(w))
        accept, alias = oracle.alias_table(w)
    w[0] = 0.0
    n = 60_000
    for u in (1, 7, 23):
        users = np.full(n, u, np.int32)
        neg = e.sample_uniform(dev(users), seed=5, offset=u * n).cpu().numpy()
        if weighted:  # and draw for draw the oracle's
            assert np.array_equal(neg, oracle.sample_weighted(indptr, indices, I, users, 5, u * n,
                                                              accept, alias))
        else:
            assert np.array_equal(neg, oracle.sample_uniform(indptr, indices, I, users, 5, u * n))
        expect = w.copy()
        expect[rows[u]] = 0.0
        expect /= expect.sum()
        ok, info = chi2_ok(np.bincount(neg, minlength=I), expect, n)
        assert ok, (u, info)


def test_weighted_draws_inside_the_training_kernels():
    """The same weights reach the fused kernels (STREAM, batched STREAM, the STRICT loop)."""
    P, Q, rows, indptr, indices = small_problem(U=300, I=200, seed=2)
    U, I = P.shape[0], Q.shape[0]
    rng = np.random.default_rng(3)
    w = rng.integers(1, 300, I).astype(np.float64) ** 0.8
    accept, alias = oracle.alias_table(w)
    users = np.sort(rng.integers(1, U, 4000)).astype(np.int32)
    pos = rng.integers(1, I, 4000).astype(np.int32)
    want = oracle.sample_weighted(indptr, indices, I, users, 8, 100, accept, alias)
    for mode in ("stream", "batched", "strict"):
        e = make_engine(P, Q)
        e.bind_seen_csr(dev(indptr), dev(indices))
        e.bind_item_weights(torch.from_numpy(w))
        e.set_optimizer(kind=0, lr=0.0)
        neg = torch.zeros(4000, dtype=torch.int32, device="cuda")
        if mode == "stream":
            e.set_stream_opts(True, 8)
            e.train_stream(dev(users), dev(pos), sampler=1, neg=neg, seed=8, offset=100)
        elif mode == "batched":
            e.train_stream_batched(dev(users), dev(pos), 256, sampler=1, neg=neg, seed=8, offset=100)
        else:
            lp, ln, sc, neg = e.step(dev(users), dev(pos), sampler=1, seed=8, offset=100)
        assert np.array_equal(neg.cpu().numpy(), want), mode


def test_adaptive_draws_follow_the_reference_formulae():
    P, Q, rows, indptr, indices = small_problem(U=40, I=90, d=32, seed=4)
    U, I, d = P.shape[0], Q.shape[0], P.shape[1]
    e = make_engine(P, Q)
    e.bind_seen_csr(dev(indptr), dev(indices))
    e.adaptive_refresh()
    sigma = Q[1:].astype(np.float64).std(axis=0, ddof=1)  # features["item"][1:].std(dim=0)
    p_geo, n = 0.08, 120_000
    for u in (3, 11, 29):
        users = np.full(n, u, np.int32)
        neg, fac, rnk = e.sample_adaptive(dev(users), p_geo, seed=2, offset=u * n, return_draws=True)
        neg, fac, rnk = neg.cpu().numpy(), fac.cpu().numpy(), rnk.cpu().numpy()
        # factor ~ |p_uf| sigma_f (torch.multinomial over unnormalised weights)
        wf = np.abs(P[u].astype(np.float64)) * sigma
        ok, info = chi2_ok(np.bincount(fac, minlength=d), wf / wf.sum(), n)
        assert ok, ("factor", u, info)
        # rank: r ~ Geometric(p) clamped to #unseen; r-1 from the top if p_uf > 0 else #unseen - r
        n_unseen = I - 1 - len(rows[u])
        pr = p_geo * (1 - p_geo) ** np.arange(n_unseen)  # P(r = k+1), k = 0 ..
        pr[-1] += (1 - p_geo) ** n_unseen  # the clamp folds the tail onto r = #unseen
        for f in np.argsort(-wf)[:3]:  # the three most frequent factors: enough draws each
            sel = fac == f
            top = P[u, f] > 0
            r_minus_1 = rnk[sel] if top else n_unseen - 1 - rnk[sel]
            ok, info = chi2_ok(np.bincount(r_minus_1, minlength=n_unseen), pr, int(sel.sum()))
            assert ok, ("rank", u, f, info)
            # the item IS the rank-th unseen of the snapshot column, literally (:109-121)
            col = Q[:, f].astype(np.float64).copy()
            col[rows[u]] = -1e13
            col[0] = -1e13
            order = np.argsort(-col, kind="stable")
            assert np.array_equal(neg[sel], order[rnk[sel]])
