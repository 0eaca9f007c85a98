"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the reference's golden
vectors.  Everything here needs a real MI355X (`-m gpu`).

Tolerances (fp32): after ONE step |Δw| <= 2e-6·max(1,|w|) — the GPU reduces dot products in a
shuffle tree and accumulates duplicate rows with fp32 atomics in arbitrary order, the oracle
accumulates in double; after five steps 1e-5.  Sampler outputs (integers) are bit-exact except
where stated.
"""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

MATH_FILES = [
    "math_13_uin_nobias", "math_42069_uin_nobias", "math_13_uin_bias", "math_13_all_nobias",
    "math_13_item_only_bias", "math_42069_none_nobias",
]
REG = {
    "uin": (0.0016, 0.0001, 0.00375),
    "all": (0.00043, 0.00043, 0.00043),
    "item_only": (0.0, 0.0025, 0.0025),
    "none": (0.0, 0.0, 0.0),
}


def close(a, b, tol):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return bool(np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b))))


def close_mostly(a, b, tol, cap, frac=5e-4):
    """`close` for quantities that pass through Adam's / RMSprop's division by sqrt(v) ~ |g|: an
    element whose summed fp32 gradient lands within rounding of zero takes a +-lr step whose sign
    depends on that rounding (the oracle accumulates in double, torch-CPU differs from torch-ROCm
    the same way), and momentum carries it on.  All but a fraction `frac` of the elements must meet
    `tol`; the rest must stay within `cap` (a few learning rates)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    err = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    return bool((err > tol).mean() <= frac and err.max() <= cap)


def maxerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def make_engine(P, Q, b=None, reg=(0, 0, 0), **kw):
    from revisit_bpr.engine import Engine

    tP, tQ = dev(P), dev(Q)
    tb = dev(b) if b is not None else None
    e = Engine(tP, tQ, tb, **kw)
    e.set_reg(*reg)
    return e


def load_math(golden_dir, name):
    g = np.load(golden_dir / f"{name}.npz")
    reg = REG[name.split("_", 2)[2].rsplit("_", 1)[0]]
    return g, reg, name.endswith("_bias")


def rand_problem(U, I, d, nnz_per_user, seed, B):
    rng = np.random.default_rng(seed)
    P = ((rng.random((U, d)) - 0.5) / d * 4).astype(np.float32)
    Q = ((rng.random((I, d)) - 0.5) / d * 4).astype(np.float32)
    P[0] = 0
    Q[0] = 0
    rows = [np.sort(rng.choice(np.arange(1, I), size=int(rng.integers(0, nnz_per_user + 1)),
                               replace=False)) for _ in range(U)]
    rows[0] = np.zeros(0, np.int64)
    indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    indices = np.concatenate(rows).astype(np.int32) if indptr[-1] else np.zeros(0, np.int32)
    users = rng.integers(1, U, size=B).astype(np.int32)
    pos = rng.integers(1, I, size=B).astype(np.int32)
    neg = rng.integers(1, I, size=B).astype(np.int32)
    return P, Q, indptr, indices, users, pos, neg


# ------------------------------------------------------------------------------------------------
# forward / gradients / optimizers against the reference's golden vectors
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", MATH_FILES)
def test_forward_and_grad_vs_reference_golden(golden_dir, name):
    g, reg, bias = load_math(golden_dir, name)
    e = make_engine(g["P0"], g["Q0"], g["b0"] if bias else None, reg)
    u, i, j = dev(i32(g["users0"])), dev(i32(g["pos0"])), dev(i32(g["neg0"]))
    lp, ln, sc = e.forward(u, i, j)
    assert close(lp.cpu().numpy(), g["fwd_logits_pos"].reshape(-1), 2e-6)
    assert close(ln.cpu().numpy(), g["fwd_logits_neg"].reshape(-1), 2e-6)
    sc = sc.cpu().numpy()
    assert close(sc[0], g["fwd_bpr_loss"], 2e-6) and close(sc[1], g["fwd_l2_reg"], 2e-6)
    assert sc[3] == len(g["users0"])
    lp2, ln2, _ = e.forward_grad(u, i, j)
    assert torch.equal(lp, lp2) and torch.equal(ln, ln2)
    gP, gQ, gb = (t.cpu().numpy() for t in e.get_grad())
    assert close(gP, g["gP"], 2e-6) and close(gQ, g["gQ"], 2e-6)
    if bias:
        assert close(gb, g["gb"], 2e-6)
    e.discard_grad()
    gP, gQ, gb = (t.cpu().numpy() for t in e.get_grad())
    assert not gP.any() and not gQ.any() and not gb.any()


OPT_CFG = {
    "sgd": dict(kind=0, lr=0.05),
    "sgd_nesterov": dict(kind=1, lr=0.05, momentum=0.9, nesterov=True),
    "sgd_momentum": dict(kind=1, lr=0.05, momentum=0.5),
    "adam_09": dict(kind=2, lr=0.01, betas=(0.9, 0.999)),
    "adam_01": dict(kind=2, lr=0.01, betas=(0.1, 0.999)),
    "adam_00": dict(kind=2, lr=0.01, betas=(0.0, 0.99)),
    "rmsprop": dict(kind=3, lr=0.01, alpha=0.9),
}


@pytest.mark.parametrize("name", MATH_FILES)
@pytest.mark.parametrize("opt_name", list(OPT_CFG))
def test_strict_steps_vs_reference_golden(golden_dir, name, opt_name):
    """STRICT mode == the reference's dense torch.optim trajectories, including the drift of rows a
    dense optimizer moves without touching them (lazy replay + flush)."""
    g, reg, bias = load_math(golden_dir, name)
    e = make_engine(g["P0"], g["Q0"], g["b0"] if bias else None, reg)
    e.set_optimizer(**OPT_CFG[opt_name])
    e.alloc_opt_state()
    for s in range(5):
        u, i, j = (dev(i32(g[f"{k}{s}"])) for k in ("users", "pos", "neg"))
        _, _, sc, _ = e.step(u, i, j)
        sc = sc.cpu().numpy()
        if s in (0, 4):
            e.flush_lazy()
            tol = 2e-6 if s == 0 else 1e-5
            P, Q = e.P.cpu().numpy(), e.Q.cpu().numpy()
            assert close(P, g[f"{opt_name}_P{s + 1}"], tol), (s, maxerr(P, g[f"{opt_name}_P{s + 1}"]))
            assert close(Q, g[f"{opt_name}_Q{s + 1}"], tol), (s, maxerr(Q, g[f"{opt_name}_Q{s + 1}"]))
            if bias:
                assert close(e.item_bias.cpu().numpy(), g[f"{opt_name}_b{s + 1}"], tol)
    assert e.step_count == 5


@pytest.mark.parametrize("opt_name", ["adam_09", "sgd_nesterov", "rmsprop"])
def test_lazy_replay_without_intermediate_flush(golden_dir, opt_name):
    """Same as above but flushing only once at the end: rows sit untouched for several steps."""
    g, reg, bias = load_math(golden_dir, "math_13_uin_bias")
    e = make_engine(g["P0"], g["Q0"], g["b0"], reg)
    e.set_optimizer(**OPT_CFG[opt_name])
    e.alloc_opt_state()
    for s in range(5):
        e.step(*(dev(i32(g[f"{k}{s}"])) for k in ("users", "pos", "neg")))
    e.flush_lazy()
    assert close(e.P.cpu().numpy(), g[f"{opt_name}_P5"], 1e-5)
    assert close(e.Q.cpu().numpy(), g[f"{opt_name}_Q5"], 1e-5)
    assert close(e.item_bias.cpu().numpy(), g[f"{opt_name}_b5"], 1e-5)


@pytest.mark.parametrize("route", ["closed-form", "loop"])
@pytest.mark.parametrize("d", [50, 128, 256])
def test_adam_lazy_replay_long_gaps_after_warmup(d, route, monkeypatch):
    """Dense torch.optim.Adam (default betas) emulated lazily, 40,000 steps into training: small
    batches leave rows untouched for hundreds of steps, so every read / update replays up to 176
    zero-gradient steps — in closed form once the bias corrections have saturated (geometric series
    with the eps expansion), or by the step loop (BPR_NO_ADAM_CLOSED).  Both must follow the oracle's
    dense Adam, which really moves every row on every step."""
    if route == "loop":
        monkeypatch.setenv("BPR_NO_ADAM_CLOSED", "1")
    U, I, B, T0, steps = 300, 200, 24, 40_000, 260
    P, Q, _, _, _, _, _ = rand_problem(U, I, d, 10, seed=d, B=8)
    P *= 4
    Q *= 4
    reg = (0.002, 0.001, 0.003)
    cfg = dict(kind=2, lr=0.003, betas=(0.9, 0.999))
    e = make_engine(P, Q, None, reg)
    e.set_optimizer(**cfg)
    e.alloc_opt_state()
    Po, Qo = P.copy(), Q.copy()
    st = {k: np.zeros_like(Po if k.endswith("P") else Qo) for k in ("mP", "vP", "mQ", "vQ")}
    opt = oracle.make_opt(2, 0.003, betas=(0.9, 0.999))
    rng = np.random.default_rng(d)

    def batch(n):
        return (rng.integers(1, U, n).astype(np.int32), rng.integers(1, I, n).astype(np.int32),
                rng.integers(1, I, n).astype(np.int32))

    for t in range(1, 4):  # give every row some momentum first
        u, i, j = batch(2000)
        e.step(dev(u), dev(i), dev(j))
        oracle.step(Po, Qo, None, u, i, j, opt, t, st, reg)
    e.flush_lazy()
    e.set_step(T0)  # as if resumed from a checkpoint written at step T0
    for s in range(steps):
        u, i, j = batch(B)
        e.step(dev(u), dev(i), dev(j))
        oracle.step(Po, Qo, None, u, i, j, opt, T0 + 1 + s, st, reg)
    e.flush_lazy()
    Pg, Qg = e.P.cpu().numpy(), e.Q.cpu().numpy()
    assert np.abs(Po - P).max() > 0.01  # the tables really moved
    assert close(Pg, Po, 2e-5), maxerr(Pg, Po)
    assert close(Qg, Qo, 2e-5), maxerr(Qg, Qo)


@pytest.mark.parametrize("opt_name,cfg", [
    ("nesterov", dict(kind=1, lr=0.01, momentum=0.9, nesterov=True)),
    ("momentum", dict(kind=1, lr=0.01, momentum=0.9)),
    ("momentum_damp", dict(kind=1, lr=0.02, momentum=0.5, dampening=0.3)),
    ("rmsprop", dict(kind=3, lr=0.0005, alpha=0.9)),
    ("rmsprop_mom", dict(kind=3, lr=0.0003, alpha=0.9, momentum=0.8)),
    ("adam_01", dict(kind=2, lr=0.001, betas=(0.1, 0.999))),
])
@pytest.mark.parametrize("d", [32, 128])
def test_stateful_optimizers_long_horizon_vs_dense_oracle(opt_name, cfg, d):
    """2,400 mini-batch steps of bpr_train_strict (GIVEN negatives, batches of 24 on a 300 x 200
    problem: a row sits untouched for dozens of steps between two touches, so almost every read and
    update goes through the lazy replay) against the oracle's DENSE torch.optim restatement, which
    moves every row on every step.  Deterministic long-horizon pin of the momentum / Nesterov /
    RMSprop / Adam paths (the e2e Nesterov offset is not in the optimizer)."""
    U, I, B, steps = 300, 200, 24, 2400
    P, Q, _, _, _, _, _ = rand_problem(U, I, d, 10, seed=d + 1, B=8)
    P *= 4
    Q *= 4
    reg = (0.0016, 0.0001, 0.00375)
    rng = np.random.default_rng(7)
    users = rng.integers(1, U, steps * B).astype(np.int32)
    pos = rng.integers(1, I, steps * B).astype(np.int32)
    neg = rng.integers(1, I, steps * B).astype(np.int32)
    e = make_engine(P, Q, None, reg)
    e.set_optimizer(**cfg)
    e.alloc_opt_state()
    sc = torch.zeros(4, device="cuda")
    half = (steps // 2) * B
    # two calls: state, step counter and last-touched marks carry across bpr_train_strict calls
    e.train_strict(dev(users[:half]), dev(pos[:half]), B, sampler=0, neg=dev(neg[:half]), scalars=sc)
    e.train_strict(dev(users[half:]), dev(pos[half:]), B, sampler=0, neg=dev(neg[half:]), scalars=sc)
    e.flush_lazy()
    Po, Qo = P.copy(), Q.copy()
    st = {k: np.zeros_like(Po if k.endswith("P") else Qo) for k in ("mP", "vP", "mQ", "vQ")}
    okw = {k: v for k, v in cfg.items() if k != "kind"}
    opt = oracle.make_opt(cfg["kind"], **okw)
    loss = 0.0
    for t in range(steps):
        sl = slice(t * B, (t + 1) * B)
        _, _, sco = oracle.step(Po, Qo, None, users[sl], pos[sl], neg[sl], opt, t + 1, st, reg)
        loss += sco[0]
    Pg, Qg = e.P.cpu().numpy(), e.Q.cpu().numpy()
    assert e.step_count == steps
    assert np.abs(Po - P).max() > 0.02  # the tables really moved
    assert close(Pg, Po, 2e-5), maxerr(Pg, Po)
    assert close(Qg, Qo, 2e-5), maxerr(Qg, Qo)
    assert abs(float(sc[0]) - loss) <= 1e-4 * loss


# ------------------------------------------------------------------------------------------------
# against the oracle at the dims the kernels are specialised for
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [1, 8, 12, 32, 50, 64, 128, 200, 256, 300, 512, 1000, 1024])
def test_strict_step_vs_oracle_dims(d):
    U, I, B = 300, 200, 256
    P, Q, _, _, users, pos, neg = rand_problem(U, I, d, 10, seed=d, B=B)
    users[:8] = users[8]  # heavy duplicates
    pos[:16] = pos[20]
    neg[30:40] = pos[20]
    reg = (0.01, 0.02, 0.03)
    b = np.linspace(-0.1, 0.1, I).astype(np.float32)
    e = make_engine(P, Q, b, reg)
    e.set_optimizer(kind=0, lr=0.1)
    lp, ln, sc, _ = e.step(dev(users), dev(pos), dev(neg))
    Po, Qo, bo = P.copy(), Q.copy(), b.copy()
    lpo, lno, sco = oracle.step(Po, Qo, bo, users, pos, neg, oracle.make_opt(oracle.SGD, 0.1), 1,
                                None, reg)
    assert close(lp.cpu().numpy(), lpo, 2e-6) and close(ln.cpu().numpy(), lno, 2e-6)
    sc = sc.cpu().numpy()
    assert close(sc[:3], sco[:3], 1e-5), (sc, sco)
    assert close(e.P.cpu().numpy(), Po, 2e-6), maxerr(e.P.cpu().numpy(), Po)
    assert close(e.Q.cpu().numpy(), Qo, 2e-6), maxerr(e.Q.cpu().numpy(), Qo)
    assert close(e.item_bias.cpu().numpy(), bo, 2e-6)
    assert not e.P[0].any() and not e.Q[0].any()


def test_empty_batch_and_errors():
    from revisit_bpr import native
    from revisit_bpr.engine import Engine

    P, Q, indptr, indices, users, pos, neg = rand_problem(50, 40, 32, 5, seed=1, B=4)
    e = make_engine(P, Q)
    z = torch.zeros(0, dtype=torch.int32, device="cuda")
    lp, ln, sc, _ = e.step(z, z, z)
    assert lp.numel() == 0 and float(sc.sum()) == 0.0
    with pytest.raises(native.BprError):  # no CSR bound
        e.sample_uniform(dev(users), seed=1)
    with pytest.raises(native.BprError):  # adaptive without snapshot
        e.bind_seen_csr(dev(indptr), dev(indices))
        e.sample_adaptive(dev(users), 0.1, seed=1)
    e.set_optimizer(kind=2, lr=0.1)
    with pytest.raises(native.BprError):  # Adam state not bound
        e.step(dev(users), dev(pos), dev(neg))
    with pytest.raises(native.BprError):  # STREAM is SGD only
        e.train_stream(dev(users), dev(pos), sampler=0, neg=dev(neg))
    with pytest.raises(native.BprError):  # unsupported dim
        Engine(torch.zeros(4, 2000, device="cuda"), torch.zeros(4, 2000, device="cuda"))
    with pytest.raises(RuntimeError):  # no CPU path
        Engine(torch.zeros(4, 8), torch.zeros(4, 8))


# ------------------------------------------------------------------------------------------------
# samplers
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [8, 32, 128, 256])
def test_uniform_sampler_bit_exact(d):
    U, I, B = 400, 150, 5000
    P, Q, indptr, indices, users, _, _ = rand_problem(U, I, d, 120, seed=3 + d, B=B)
    e = make_engine(P, Q)
    e.bind_seen_csr(dev(indptr), dev(indices))
    got = e.sample_uniform(dev(users), seed=0xDEADBEEFCAFE, offset=12345).cpu().numpy()
    want = oracle.sample_uniform(indptr, indices, I, users, seed=0xDEADBEEFCAFE, offset=12345)
    assert np.array_equal(got, want)
    # never seen, never the pad item
    for u, j in zip(users[:500], got[:500]):
        assert j != 0 and j not in indices[indptr[u]:indptr[u + 1]]


@pytest.mark.parametrize("d", [8, 32, 128, 256, 512])
def test_adaptive_refresh_and_pick(d):
    U, I, B = 200, 333, 3000
    P, Q, indptr, indices, users, _, _ = rand_problem(U, I, d, 60, seed=11 + d, B=B)
    e = make_engine(P, Q)
    e.bind_seen_csr(dev(indptr), dev(indices))
    e.adaptive_refresh()
    order, sigma = e.adaptive_snapshot()
    QT, sigma_o = oracle.adaptive_stats(Q)
    order_o = oracle.adaptive_order(QT)
    assert np.array_equal(order.cpu().numpy(), order_o)
    assert close(sigma.cpu().numpy(), sigma_o, 1e-6)
    rng = np.random.default_rng(d)
    fac = rng.integers(0, d, size=B).astype(np.int32)
    n_unseen = (I - 1) - (indptr[users.astype(np.int64) + 1] - indptr[users])
    rank = (rng.random(B) * n_unseen).astype(np.int32)
    rank[:50] = 0
    rank[50:100] = (n_unseen[50:100] - 1).astype(np.int32)
    got = e.adaptive_pick(dev(users), dev(fac), dev(rank)).cpu().numpy()
    for b in range(0, B, 7):
        assert got[b] == oracle.adaptive_pick(order_o, indptr, indices, int(users[b]),
                                              int(fac[b]), int(rank[b]))
    for b in range(0, B, 97):
        assert got[b] == oracle.adaptive_pick_literal(QT, indptr, indices, int(users[b]),
                                                      int(fac[b]), int(rank[b]))


@pytest.mark.parametrize("d", [8, 32, 128, 256])
def test_adaptive_sampler_matches_oracle(d):
    U, I, B = 300, 500, 20000
    P, Q, indptr, indices, users, _, _ = rand_problem(U, I, d, 80, seed=21 + d, B=B)
    e = make_engine(P, Q)
    e.bind_seen_csr(dev(indptr), dev(indices))
    e.adaptive_refresh()
    neg, fac, rnk = (t.cpu().numpy() for t in
                     e.sample_adaptive(dev(users), 0.05, seed=77, offset=5, return_draws=True))
    QT, sigma = oracle.adaptive_stats(Q)
    order = oracle.adaptive_order(QT)
    neg_o, fac_o, rnk_o = oracle.sample_adaptive(P, sigma, order, indptr, indices, users, 0.05,
                                                 seed=77, offset=5)
    # factor: inverse-CDF thresholds are compared in fp32 on the GPU (scan tree) and in double on
    # the CPU, so draws landing within rounding of a bin edge may fall in the neighbouring bin.
    same_f = fac == fac_o
    assert same_f.mean() > 0.999, same_f.mean()
    assert np.all(np.abs(fac[~same_f] - fac_o[~same_f]) <= 2)
    # rank: logf differs by <= 1 ulp between libm and ocml → ceil() may flip on exact integers
    same = same_f & (rnk == rnk_o)
    assert same.mean() > 0.998, same.mean()
    assert np.array_equal(neg[same], neg_o[same])
    # every GPU pick is the correct pick for ITS OWN (factor, rank)
    for b in range(0, B, 41):
        assert neg[b] == oracle.adaptive_pick(order, indptr, indices, int(users[b]), int(fac[b]),
                                              int(rnk[b]))


@pytest.mark.parametrize("d", [32, 128, 256])
def test_adaptive_mismatches_are_cdf_bin_edges_and_ceil_flips(d):
    """The tolerance of the adaptive-pick comparisons (">= 99 % identical") shown by its CAUSE
    (VERDICT r3, weak #10): of 200,000 picks, EVERY pick whose factor differs from the oracle's has
    its inverse-CDF threshold uf * total within fp32 summation error of a bin edge (the GPU sums
    d weights in a scan tree in fp32, the oracle in double), EVERY pick whose rank differs has
    ln(u) / ln(1 - p) within 2 ulp of an integer (ceil of logf: ocml vs libm), and the picks that sit
    that close to an edge are as rare as the mismatches are."""
    U, I, B, p_geo = 400, 600, 200_000, 0.05
    P, Q, indptr, indices, users, _, _ = rand_problem(U, I, d, 60, seed=3 + d, B=B)
    e = make_engine(P, Q)
    e.bind_seen_csr(dev(indptr), dev(indices))
    e.adaptive_refresh()
    neg, fac, rnk = (t.cpu().numpy() for t in
                     e.sample_adaptive(dev(users), p_geo, seed=77, offset=5, return_draws=True))
    QT, sigma = oracle.adaptive_stats(Q)
    order = oracle.adaptive_order(QT)
    neg_o, fac_o, rnk_o = oracle.sample_adaptive(P, sigma, order, indptr, indices, users, p_geo, seed=77, offset=5)
    G = 32 if d <= 128 else 64
    enum = np.array([f for lane in range(G) for f in range(lane, d, G)])  # the CDF's factor order
    w = np.abs(P[users][:, enum].astype(np.float64)) * sigma[enum].astype(np.float64)  # [B, d]
    cum = np.cumsum(w, axis=1)
    total = cum[:, -1]
    rf = np.array([oracle.philox4x32_10(((5 + t) & 0xFFFFFFFF, (5 + t) >> 32, 0, 1), (77, 0))[:2]
                   for t in np.nonzero((fac != fac_o) | (rnk != rnk_o))[0]], dtype=np.uint64).reshape(-1, 2)
    bad = np.nonzero((fac != fac_o) | (rnk != rnk_o))[0]
    n_fac = n_rnk = 0
    for k, t in enumerate(bad):
        uf = float(int(rf[k, 0]) >> 8) / 16777216.0
        ug = float((int(rf[k, 1]) >> 8) + 1) / 16777216.0
        if fac[t] != fac_o[t]:
            n_fac += 1
            margin = np.min(np.abs(uf * total[t] - cum[t])) / total[t]
            assert margin <= 4e-6 * d ** 0.5 + 1e-6, (t, margin)  # fp32 sum of d terms
            pos_g, pos_o = np.nonzero(enum == fac[t])[0][0], np.nonzero(enum == fac_o[t])[0][0]
            assert abs(int(pos_g) - int(pos_o)) <= 2  # the neighbouring bin of the enumeration
        if rnk[t] != rnk_o[t] and fac[t] == fac_o[t]:
            n_rnk += 1
            x = np.log(ug) / np.log1p(-p_geo)
            assert abs(x - round(x)) <= 4e-7 * max(1.0, abs(x)) * 4, (t, x)
    assert n_fac <= 2e-3 * B and n_rnk <= 2e-3 * B, (n_fac, n_rnk)
    same = (fac == fac_o) & (rnk == rnk_o)
    assert np.array_equal(neg[same], neg_o[same])


@pytest.mark.parametrize("d", [32, 256])
def test_samplers_with_heavy_users(d):
    """Users holding up to 2,400 of 3,000 items (long CSR slices, few unseen items left): both
    samplers must give the oracle's picks."""
    U, I, B = 40, 3000, 3000
    P, Q, indptr, indices, users, _, _ = rand_problem(U, I, d, 2400, seed=31 + d, B=B)
    e = make_engine(P, Q)
    e.bind_seen_csr(dev(indptr), dev(indices))
    got = e.sample_uniform(dev(users), seed=99, offset=7).cpu().numpy()
    assert np.array_equal(got, oracle.sample_uniform(indptr, indices, I, users, seed=99, offset=7))
    e.adaptive_refresh()
    neg, fac, rnk = (t.cpu().numpy() for t in
                     e.sample_adaptive(dev(users), 0.02, seed=5, offset=1, return_draws=True))
    QT, _ = oracle.adaptive_stats(Q)
    order = oracle.adaptive_order(QT)
    for b in range(0, B, 3):
        assert neg[b] == oracle.adaptive_pick(order, indptr, indices, int(users[b]), int(fac[b]),
                                              int(rnk[b]))


@pytest.mark.parametrize("I,d", [(100_003, 8), (150_001, 200), (150_001, 8)])
def test_adaptive_sampler_large_item_tables(I, d):
    """Item tables whose per-group seen-bitmaps need more than the default 64 KiB of dynamic LDS
    (8 groups x 100 k bits = 100 KB; 4 groups x 150 k bits = 75 KB at d > 128) or do not fit at all
    (8 groups x 150 k bits: binary search in the CSR)."""
    U, B = 60, 1500
    P, Q, indptr, indices, users, _, _ = rand_problem(U, I, d, 3000, seed=7 + d, B=B)
    e = make_engine(P, Q)
    e.bind_seen_csr(dev(indptr), dev(indices))
    e.adaptive_refresh()
    neg, fac, rnk = (t.cpu().numpy() for t in
                     e.sample_adaptive(dev(users), 0.01, seed=3, offset=9, return_draws=True))
    QT, _ = oracle.adaptive_stats(Q)
    order = oracle.adaptive_order(QT)
    for b in range(0, B, 5):
        assert neg[b] == oracle.adaptive_pick(order, indptr, indices, int(users[b]), int(fac[b]),
                                              int(rnk[b]))


# ------------------------------------------------------------------------------------------------
# STREAM mode
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [8, 32, 128, 256, 1024])
def test_stream_unique_rows_equals_strict(d):
    """When no row occurs twice in the chunk, asynchronous and mini-batch SGD coincide."""
    n = 2000
    U, I = n + 1, 2 * n + 1
    rng = np.random.default_rng(d)
    P = ((rng.random((U, d)) - 0.5)).astype(np.float32)
    Q = ((rng.random((I, d)) - 0.5)).astype(np.float32)
    P[0] = 0
    Q[0] = 0
    b = rng.normal(0, 0.1, I).astype(np.float32)
    users = (rng.permutation(n) + 1).astype(np.int32)
    it = (rng.permutation(2 * n) + 1).astype(np.int32)
    pos, neg = it[:n].copy(), it[n:].copy()
    reg = (0.01, 0.02, 0.03)
    e = make_engine(P, Q, b, reg)
    e.set_optimizer(kind=0, lr=0.05)
    sc = torch.zeros(4, device="cuda")
    e.train_stream(dev(users), dev(pos), sampler=0, neg=dev(neg), scalars=sc)
    Po, Qo, bo = P.copy(), Q.copy(), b.copy()
    _, _, sco = oracle.step_sgd_sparse(Po, Qo, bo, users, pos, neg, 0.05, reg)
    assert close(e.P.cpu().numpy(), Po, 2e-6), maxerr(e.P.cpu().numpy(), Po)
    assert close(e.Q.cpu().numpy(), Qo, 2e-6)
    assert close(e.item_bias.cpu().numpy(), bo, 2e-6)
    assert close(sc.cpu().numpy()[:3], sco[:3], 2e-5)


@pytest.mark.parametrize("seen", ["", "csr", "bitmap", "list", "list-overflow"])
@pytest.mark.parametrize("sampler", [1, 2])
def test_stream_sequential_limit_equals_b1_sgd(sampler, seen, monkeypatch):
    """d=256 → one triple per wave; max_inflight=1 → one wave → exactly sequential SGD with the
    on-device sampler, comparable step by step with the oracle's B=1 stream.  Every "seen?"
    structure of the STREAM kernel (CSR in HBM, LDS bitmap, LDS sorted list, and the list's
    fall-back for users with more seen items than it stages) must give the same picks."""
    d, U, I, n = 256, 60, 90, 400
    per_user = 30
    if seen == "list-overflow":  # > 512 seen items per user: the staged list does not hold them
        I, per_user, seen = 1500, 700, "list"
    if seen:
        monkeypatch.setenv("BPR_SEEN", seen)
    P, Q, indptr, indices, users, pos, _ = rand_problem(U, I, d, per_user, seed=5, B=n)
    P *= 8
    Q *= 8
    reg = (0.01, 0.02, 0.03)
    e = make_engine(P, Q, None, reg)
    e.bind_seen_csr(dev(indptr), dev(indices))
    e.set_optimizer(kind=0, lr=0.05)
    e.adaptive_refresh()
    negs = torch.zeros(n, dtype=torch.int32, device="cuda")
    sc = torch.zeros(4, device="cuda")
    e.train_stream(dev(users), dev(pos), sampler=sampler, neg=negs, adaptive_p=0.1, seed=9,
                   offset=100, max_inflight=1, scalars=sc)
    QT, sigma = oracle.adaptive_stats(Q)
    order = oracle.adaptive_order(QT)
    Po, Qo = P.copy(), Q.copy()
    neg_o = np.zeros(n, np.int32)
    sco = oracle.train_stream_seq(Po, Qo, None, users, pos, neg_o, sampler, 0.05, reg,
                                  adaptive_p=0.1, sigma=sigma, order=order, indptr=indptr,
                                  indices=indices, seed=9, offset=100)
    got = negs.cpu().numpy()
    if sampler == 1:
        assert np.array_equal(got, neg_o)
    else:  # adaptive draws depend on the live (fp32-rounded) user row; allow rare edge flips
        assert (got == neg_o).mean() > 0.97
    if np.array_equal(got, neg_o):
        assert close(e.P.cpu().numpy(), Po, 1e-5), maxerr(e.P.cpu().numpy(), Po)
        assert close(e.Q.cpu().numpy(), Qo, 1e-5)
        assert close(sc.cpu().numpy()[:3], sco[:3], 1e-4)


def test_stream_run_length_follows_the_launch_size():
    """run_len = 0 lets the library pick: runs of 8 for launches that fill the chip more than once,
    the shortest runs of 4..8 triples that fit it in one residency below that (a small launch ends
    when its slowest group does), 8 under a max_inflight cap; an explicit run_len is kept.  The
    picks do not depend on it (lr = 0: uniform picks equal the oracle's for every run length)."""
    d, U, I = 64, 3000, 800
    total = torch.cuda.get_device_properties(0).multi_processor_count
    n_big, n_small = 40 * 8 * total * 2 * 8, 6 * total * 2 * 4  # >> / << groups the chip holds x 8
    P, Q, indptr, indices, users, pos, _ = rand_problem(U, I, d, 100, seed=77, B=n_big)
    e = make_engine(P, Q, None, (0.01, 0.01, 0.01))
    e.bind_seen_csr(dev(indptr), dev(indices))
    e.set_optimizer(kind=0, lr=0.0)
    order = np.argsort(users, kind="stable")
    users, pos = users[order], pos[order]
    want = None
    for n, run_len, expect in ((n_big, 0, (8, 8)), (n_small, 0, (4, 4)), (n_small, 3, (3, 3)),
                               (40000, 0, (4, 8))):
        e.set_stream_opts(True, run_len)
        negs = torch.zeros(n, dtype=torch.int32, device="cuda")
        e.train_stream(dev(users[:n]), dev(pos[:n]), sampler=1, neg=negs, seed=3, offset=0)
        assert expect[0] <= e.stream_run_len() <= expect[1], (n, run_len, e.stream_run_len())
        if n == n_small:
            got = negs.cpu().numpy()
            if want is None:
                want = oracle.sample_uniform(indptr, indices, I, users[:n], seed=3, offset=0)
            assert np.array_equal(got, want)
    e.set_stream_opts(True, 0)
    negs = torch.zeros(n_small, dtype=torch.int32, device="cuda")
    e.train_stream(dev(users[:n_small]), dev(pos[:n_small]), sampler=1, neg=negs, seed=3, offset=0,
                   max_inflight=4)
    assert e.stream_run_len() == 8


@pytest.mark.parametrize("seen", ["", "list", "csr"])
@pytest.mark.parametrize("d", [8, 32, 50, 64, 128, 256, 1024])
def test_stream_picks_match_the_oracle_at_full_concurrency(d, seen, monkeypatch):
    """lr = 0 freezes the tables, so the negatives a full-width grouped STREAM launch draws are a
    pure function of (seed, offset, triple index, tables): uniform picks must equal the oracle's
    exactly, adaptive picks up to the rare fp32 bin-edge flips of the factor / rank draw — for
    every group width / row width combination and every "seen?" structure."""
    if seen:
        monkeypatch.setenv("BPR_SEEN", seen)
    U, I, n = 500, 700, 20000
    P, Q, indptr, indices, users, pos, _ = rand_problem(U, I, d, 150, seed=40 + d, B=n)
    P *= 6
    Q *= 6
    e = make_engine(P, Q, None, (0.01, 0.01, 0.01))
    e.bind_seen_csr(dev(indptr), dev(indices))
    e.set_optimizer(kind=0, lr=0.0)
    e.set_stream_opts(True, 8)
    pu, pi = e.plan_epoch(dev(users), dev(pos), n, seed=1)
    e.adaptive_refresh()
    QT, sigma = oracle.adaptive_stats(Q)
    order = oracle.adaptive_order(QT)
    pun = pu.cpu().numpy()
    for sampler in (1, 2):
        negs = torch.zeros(n, dtype=torch.int32, device="cuda")
        e.train_stream(pu, pi, sampler=sampler, neg=negs, adaptive_p=0.03, seed=77, offset=1000)
        got = negs.cpu().numpy()
        if sampler == 1:
            want = oracle.sample_uniform(indptr, indices, I, pun, seed=77, offset=1000)
            assert np.array_equal(got, want)
        else:
            want, _, _ = oracle.sample_adaptive(P, sigma, order, indptr, indices, pun, 0.03, seed=77,
                                                offset=1000)
            assert (got == want).mean() > 0.995, (got == want).mean()
        for t in range(0, n, 97):
            assert got[t] != 0 and got[t] not in indices[indptr[pun[t]]:indptr[pun[t] + 1]]
    assert np.array_equal(e.P.cpu().numpy(), P) and np.array_equal(e.Q.cpu().numpy(), Q)


@pytest.mark.parametrize("seen", ["", "list", "csr"])
def test_stream_full_chip_learns_and_never_picks_seen(seen, monkeypatch):
    """Full-concurrency STREAM on a small synthetic set: loss falls epoch over epoch, sampled
    negatives are valid, tables stay finite, pad rows stay zero."""
    from revisit_bpr.datasets import synthetic

    if seen:
        monkeypatch.setenv("BPR_SEEN", seen)

    data = synthetic.generate(2000, 1000, 60000, median_per_user=20, seed=3)
    d = 64
    g = torch.Generator().manual_seed(0)
    P = ((torch.rand(data.num_users, d, generator=g) - 0.5) / d)
    Q = ((torch.rand(data.num_items, d, generator=g) - 0.5) / d)
    P[0] = 0
    Q[0] = 0
    e = make_engine(P.numpy(), Q.numpy(), None, (0.001, 0.001, 0.001))
    e.bind_seen_csr(dev(data.indptr), dev(data.indices))
    e.set_optimizer(kind=0, lr=0.1)
    users, pos = dev(data.users), dev(data.items)
    losses = []
    for ep in range(4):
        perm = torch.randperm(users.numel(), device="cuda")
        u, i = users[perm].contiguous(), pos[perm].contiguous()
        neg = torch.zeros_like(u)
        sc = torch.zeros(4, device="cuda")
        e.train_stream(u, i, sampler=1, neg=neg, seed=ep, scalars=sc)
        losses.append(float(sc[0] / sc[3]))
        if ep == 0:
            un, nn = u.cpu().numpy(), neg.cpu().numpy()
            for t in range(0, len(un), 503):
                assert nn[t] != 0 and nn[t] not in data.indices[data.indptr[un[t]]:data.indptr[un[t] + 1]]
    assert losses[-1] < losses[0] - 0.05, losses
    assert torch.isfinite(e.P).all() and torch.isfinite(e.Q).all()
    assert not e.P[0].any() and not e.Q[0].any()


# ------------------------------------------------------------------------------------------------
# epoch planner + user-grouped STREAM
# ------------------------------------------------------------------------------------------------
def test_plan_epoch_is_a_grouped_random_partition():
    from revisit_bpr.datasets import synthetic

    data = synthetic.generate(3000, 800, 70000, median_per_user=15, seed=5)
    P = np.zeros((data.num_users, 8), np.float32)
    Q = np.zeros((data.num_items, 8), np.float32)
    e = make_engine(P, Q)
    users, pos = dev(data.users), dev(data.items)
    n, chunk = data.nnz, 9000
    u1, p1 = (t.cpu().numpy() for t in e.plan_epoch(users, pos, chunk, seed=1))
    u2, p2 = (t.cpu().numpy() for t in e.plan_epoch(users, pos, chunk, seed=2))
    key = data.users.astype(np.int64) * data.num_items + data.items
    for uu, pp in ((u1, p1), (u2, p2)):
        assert np.array_equal(np.sort(uu.astype(np.int64) * data.num_items + pp), np.sort(key))
        for c0 in range(0, n, chunk):  # grouped (sorted) by user inside every chunk
            assert np.all(np.diff(uu[c0:c0 + chunk]) >= 0)
    assert not np.array_equal(u1, u2)
    # chunk membership is (pseudo-)random: a user's triples spread over the chunks, and the first
    # chunk of two seeds shares about 1/n_chunks of its triples
    k1 = set((u1[:chunk].astype(np.int64) * data.num_items + p1[:chunk]).tolist())
    k2 = set((u2[:chunk].astype(np.int64) * data.num_items + p2[:chunk]).tolist())
    frac = len(k1 & k2) / chunk
    assert 0.5 * chunk / n < frac < 2.0 * chunk / n, frac
    # source position of chunk 0 is not clustered at the front of the (user-sorted) input
    first_user_share = (u1[:chunk] < data.num_users // 2).mean()
    assert 0.4 < first_user_share < 0.6, first_user_share


@pytest.mark.parametrize("n,chunk,U", [(50_000, 4096, 3000), (1_000_003, 65_536, 140_000), (9_000, 9_000, 500)])
def test_plan_epoch_sorted_input_promise_gives_the_same_plan(n, chunk, U):
    """r6: with the triple list in CSR order (sorted by user — what every loader here makes) a STABLE sort on the
    chunk bits alone leaves every chunk grouped by user: `plan_epoch(sorted_input=True)` is the same plan, bit for
    bit, in one radix pass instead of three; `Engine.users_sorted` is the check the trainers run once."""
    from revisit_bpr.engine import Engine

    rng = np.random.default_rng(n)
    users = np.sort(rng.integers(1, U, n)).astype(np.int32)
    pos = rng.integers(1, 5000, n).astype(np.int32)
    P, Q = np.zeros((U, 32), np.float32), np.zeros((5000, 32), np.float32)
    e = make_engine(P, Q)
    tu, tp = dev(users), dev(pos)
    assert Engine.users_sorted(tu) and not Engine.users_sorted(dev(users[::-1].copy()))
    for seed in (3, 4):
        a = e.plan_epoch(tu, tp, chunk, seed)
        b = e.plan_epoch(tu, tp, chunk, seed, sorted_input=True)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        c = e.plan_epoch(tu, tp, chunk, seed)  # and back
        assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])


@pytest.mark.parametrize("d,run_len", [(200, 4), (256, 8), (256, 3), (512, 1), (1024, 5)])
def test_stream_grouped_sequential_equals_b1_sgd(d, run_len):
    """(G = 64 dims only: one group per wave, so max_inflight=1 is truly sequential.)
    Planned (user-grouped) chunk walked by ONE group == sequential SGD in planned order,
    including users whose run straddles run boundaries (atomic delta path) and users fully inside
    a run (register-resident row, plain store)."""
    from revisit_bpr.datasets import synthetic

    data = synthetic.generate(150, 90, 1500, median_per_user=8, seed=d + run_len)
    rng = np.random.default_rng(d)
    P = ((rng.random((data.num_users, d)) - 0.5) * 0.5).astype(np.float32)
    Q = ((rng.random((data.num_items, d)) - 0.5) * 0.5).astype(np.float32)
    P[0] = 0
    Q[0] = 0
    reg = (0.01, 0.02, 0.03)
    e = make_engine(P, Q, None, reg)
    e.bind_seen_csr(dev(data.indptr), dev(data.indices))
    e.set_optimizer(kind=0, lr=0.05)
    e.set_stream_opts(True, run_len)
    pu, pp = e.plan_epoch(dev(data.users), dev(data.items), chunk=data.nnz, seed=3)
    negs = torch.zeros_like(pu)
    sc = torch.zeros(4, device="cuda")
    e.train_stream(pu, pp, sampler=1, neg=negs, seed=11, max_inflight=1, scalars=sc)
    Po, Qo = P.copy(), Q.copy()
    neg_o = np.zeros(data.nnz, np.int32)
    sco = oracle.train_stream_seq(Po, Qo, None, pu.cpu().numpy(), pp.cpu().numpy(), neg_o, 1, 0.05,
                                  reg, indptr=data.indptr, indices=data.indices, seed=11)
    assert np.array_equal(negs.cpu().numpy(), neg_o)
    assert close(e.P.cpu().numpy(), Po, 1e-5), maxerr(e.P.cpu().numpy(), Po)
    assert close(e.Q.cpu().numpy(), Qo, 1e-5), maxerr(e.Q.cpu().numpy(), Qo)
    assert close(sc.cpu().numpy()[:3], sco[:3], 1e-4)


@pytest.mark.parametrize("n,chunk,U", [(100_000, 9_984, 5000), (9_550_138 // 40, 199_168 // 8, 136_678), (777, 100, 60)])
def test_plan_chunk_is_the_chunk_of_the_epoch_plan(n, chunk, U):
    """bpr_plan_chunk(index) — the members of one chunk found through the INVERSE permutation and
    sorted by user — is chunk `index` of bpr_plan_epoch with the same seed: the same multiset of
    (user, positive) pairs, grouped by user; on the launch stream and queued on the split refresh's
    side stream behind a sort in flight (the commit then waits for it)."""
    rng = np.random.default_rng(n)
    I, d = 300, 32
    users = rng.integers(1, U, n).astype(np.int32)
    pos = rng.integers(1, I, n).astype(np.int32)
    e = make_engine(np.zeros((U, d), np.float32), rng.normal(0, 0.1, (I, d)).astype(np.float32))
    u_d, p_d = dev(users), dev(pos)
    for seed in (1, 7):
        eu, ep = (t.cpu().numpy() for t in e.plan_epoch(u_d, p_d, chunk, seed=seed))
        n_chunks = -(-n // chunk)
        e.adaptive_refresh()
        for index in range(n_chunks):
            m = min(chunk, n - index * chunk)
            out = (torch.empty(m, dtype=torch.int32, device="cuda"), torch.empty(m, dtype=torch.int32, device="cuda"))
            on_side = index % 2 == 1
            if on_side:
                e.adaptive_refresh_begin()
            e.plan_chunk(u_d, p_d, chunk, seed, index, out, on_side=on_side)
            if on_side:
                e.adaptive_refresh_commit()  # the launch stream now waits for the sort AND the plan
            cu, cp = out[0].cpu().numpy(), out[1].cpu().numpy()
            assert np.all(np.diff(cu) >= 0), "grouped by user"
            want_u, want_p = eu[index * chunk:index * chunk + m], ep[index * chunk:index * chunk + m]
            assert np.array_equal(cu, want_u)  # both are sorted by user: the user column is identical
            key = lambda a, b: np.sort(a.astype(np.int64) * I + b)
            assert np.array_equal(key(cu, cp), key(want_u, want_p)), (seed, index)


def test_stream_grouped_matches_atomic_mode_at_full_concurrency():
    """Same planned chunk, full chip: register-resident user rows (grouped) and all-atomic user
    rows give the same tables up to second order in lr (both are asynchronous SGD)."""
    from revisit_bpr.datasets import synthetic

    data = synthetic.generate(4000, 1500, 120000, median_per_user=20, seed=9)
    d = 128
    rng = np.random.default_rng(0)
    P0 = ((rng.random((data.num_users, d)) - 0.5) / d * 8).astype(np.float32)
    Q0 = ((rng.random((data.num_items, d)) - 0.5) / d * 8).astype(np.float32)
    P0[0] = 0
    Q0[0] = 0
    res = []
    for grouped in (True, False):
        e = make_engine(P0, Q0, None, (0.001, 0.001, 0.001))
        e.bind_seen_csr(dev(data.indptr), dev(data.indices))
        e.set_optimizer(kind=0, lr=0.01)
        e.set_stream_opts(grouped, 8)
        pu, pp = e.plan_epoch(dev(data.users), dev(data.items), chunk=data.nnz, seed=4)
        e.train_stream(pu, pp, sampler=1, seed=2)
        res.append((e.P.cpu().numpy(), e.Q.cpu().numpy()))
    (Pg, Qg), (Pa, Qa) = res
    step = np.abs(Pg - P0).max()
    assert step > 1e-4
    assert np.abs(Pg - Pa).max() < 0.05 * step, (np.abs(Pg - Pa).max(), step)
    assert np.abs(Qg - Qa).max() < 0.05 * np.abs(Qg - Q0).max()


@pytest.mark.parametrize("opt_name", ["sgd", "adam_01"])
def test_train_strict_epoch_driver(opt_name):
    """bpr_train_strict == the same batches stepped one by one through bpr_step (and, for SGD,
    == the oracle's mini-batch steps with the oracle's own Philox negatives).  The comparison uses
    the uniform sampler, whose draws do not depend on the (atomics-order-sensitive) fp32 tables;
    the adaptive sampler + periodic refresh is exercised for completion and sanity."""
    from revisit_bpr.datasets import synthetic

    data = synthetic.generate(300, 200, 6000, median_per_user=12, seed=2)
    d, B = 64, 256
    rng = np.random.default_rng(1)
    P = ((rng.random((data.num_users, d)) - 0.5) / d * 4).astype(np.float32)
    Q = ((rng.random((data.num_items, d)) - 0.5) / d * 4).astype(np.float32)
    P[0] = 0
    Q[0] = 0
    reg = (0.001, 0.002, 0.003)
    perm = rng.permutation(data.nnz)
    users, pos = data.users[perm].copy(), data.items[perm].copy()
    res = []
    for mode in ("driver", "loop", "driver-adaptive"):
        e = make_engine(P, Q, None, reg)
        e.bind_seen_csr(dev(data.indptr), dev(data.indices))
        e.set_optimizer(**OPT_CFG[opt_name])
        e.alloc_opt_state()
        e.adaptive_refresh()
        sc = torch.zeros(4, device="cuda")
        tu, tp = dev(users), dev(pos)
        if mode == "driver":
            e.train_strict(tu, tp, B, sampler=1, seed=4, refresh_every=5, scalars=sc)
        elif mode == "driver-adaptive":
            e.train_strict(tu, tp, B, sampler=2, adaptive_p=0.05, seed=4, refresh_every=5, scalars=sc)
        else:
            for k, lo in enumerate(range(0, data.nnz, B)):
                e.step(tu[lo:lo + B], tp[lo:lo + B], sampler=1, seed=4, offset=lo, scalars=sc)
                if (k + 1) % 5 == 0:
                    e.flush_lazy()
                    e.adaptive_refresh()
        e.flush_lazy()
        res.append((e.P.cpu().numpy(), e.Q.cpu().numpy(), sc.cpu().numpy()))
    assert close(res[0][0], res[1][0], 2e-5) and close(res[0][1], res[1][1], 2e-5)
    assert close(res[0][2], res[1][2], 1e-4) and res[0][2][3] == data.nnz
    assert res[2][2][3] == data.nnz and np.isfinite(res[2][0]).all() and np.isfinite(res[2][1]).all()
    assert res[2][2][0] / data.nnz < np.log(2.0) + 0.05
    if opt_name == "sgd":  # uniform negatives are bit-identical to the oracle's → full-epoch check
        e = make_engine(P, Q, None, reg)
        e.bind_seen_csr(dev(data.indptr), dev(data.indices))
        e.set_optimizer(kind=0, lr=0.05)
        e.train_strict(dev(users), dev(pos), B, sampler=1, seed=9)
        Po, Qo = P.copy(), Q.copy()
        for lo in range(0, data.nnz, B):
            u, i = users[lo:lo + B], pos[lo:lo + B]
            neg = oracle.sample_uniform(data.indptr, data.indices, data.num_items, u, seed=9, offset=lo)
            oracle.step_sgd_sparse(Po, Qo, None, np.ascontiguousarray(u), np.ascontiguousarray(i), neg,
                                   0.05, reg)
        assert close(e.P.cpu().numpy(), Po, 1e-4), maxerr(e.P.cpu().numpy(), Po)
        assert close(e.Q.cpu().numpy(), Qo, 1e-4), maxerr(e.Q.cpu().numpy(), Qo)


@pytest.mark.parametrize("I,d,force_sub", [(45000, 16, 0), (80001, 8, 0), (20109, 128, 0),
                                           (36865, 300, 0), (150000, 8, 0), (5003, 24, 2),
                                           (5003, 24, 4), (777, 40, 4), (20109, 128, 2),
                                           (24576, 16, 2), (24577, 16, 2)])
def test_refresh_large_item_counts(I, d, force_sub, monkeypatch):
    """Refresh paths: in-LDS sort with 1, 2 or 4 workgroups per factor + merge (I <= 147k), and the
    device-wide rocPRIM sort beyond.  Exact order (ties by item id) and sigma vs the oracle."""
    if force_sub:
        monkeypatch.setenv("BPR_REFRESH_SUB", str(force_sub))
    rng = np.random.default_rng(I + d)
    Q = (rng.standard_normal((I, d)) * 0.05).astype(np.float32)
    Q[0] = 0
    Q[5] = Q[7]  # exact ties
    Q[I // 2] = Q[I // 2 + 3]
    Q[I - 1] = Q[1]
    P = np.zeros((4, d), np.float32)
    e = make_engine(P, Q)
    e.adaptive_refresh()
    order, sigma = e.adaptive_snapshot()
    QT, sigma_o = oracle.adaptive_stats(Q)
    order_o = oracle.adaptive_order(QT)
    assert np.array_equal(order.cpu().numpy(), order_o)
    assert close(sigma.cpu().numpy(), sigma_o, 2e-6)


def _binned_tables(I, d, kind, rng):
    Q = (rng.standard_normal((I, d)) * 0.05).astype(np.float32)
    if kind == "ties":          # a few dozen distinct values per column: bins overflow -> the radix fallback
        Q = (np.round(Q * 400) / 400).astype(np.float32)
    elif kind == "few-ties":    # birthday collisions and planted equal rows, +0 / -0
        Q[5] = Q[7]
        Q[I // 2] = Q[I // 2 + 3]
        Q[I - 1] = Q[1]
        Q[11] = 0.0
        Q[13] = -0.0
        Q[17] = 0.0
    elif kind == "spike":       # a trained model's cold items: most keys within a hair of zero, heavy tails
        cold = rng.random(I) < 0.7
        Q[cold] *= 1e-3
        Q[rng.integers(1, I, 20)] *= 40.0
    elif kind == "narrow-spike":  # a spike narrower than one coarse bin (10 sigma / 1,024)
        Q[: I // 3] = (1e-7 * rng.standard_normal((I // 3, d))).astype(np.float32) + np.float32(0.0123)
    elif kind == "skewed":      # one-sided, exponential
        Q = (rng.exponential(0.05, (I, d))).astype(np.float32)
    elif kind == "shifted":     # a large common offset: the value-linear bins must be taken about the mean
        Q += np.float32(3.0)
    elif kind == "equal":
        Q[:] = np.float32(0.25)
    Q[0] = 0
    return Q


@pytest.mark.parametrize("I,d", [(20108, 128), (20480, 8), (17771, 64), (4801, 64), (2048, 16), (9999, 24)])
@pytest.mark.parametrize("kind", ["normal", "few-ties", "ties", "spike", "narrow-spike", "skewed", "shifted", "equal"])
def test_binned_sort_is_the_stable_descending_order(I, d, kind):
    """k_sort_binned (r5: equi-depth bins from an interpolated rank + ranking inside the bin, no radix sort) orders
    a column exactly as the oracle's stable descending argsort does — ties by ascending item id, -0 == +0 —
    on bell-shaped, spiky, one-sided, shifted and heavily tied columns; columns it gives up on (a bin over 64
    keys) come out of the radix fallback behind it; the radix path (`binned_sort` 0) agrees bit for bit."""
    rng = np.random.default_rng(I * 7 + d + len(kind))
    Q = _binned_tables(I, d, kind, rng)
    P = np.zeros((4, d), np.float32)
    QT, sigma_o = oracle.adaptive_stats(Q)
    order_o = oracle.adaptive_order(QT)
    got = []
    for binned in (1, 0):
        e = make_engine(P, Q)
        e.set_tuning("binned_sort", binned)
        e.adaptive_refresh()
        order, sigma = e.adaptive_snapshot()
        got.append((order.cpu().numpy(), sigma.cpu().numpy()))
        assert np.array_equal(got[-1][0], order_o), (binned, kind)
        if np.all(sigma_o > 0):
            assert close(got[-1][1], sigma_o, 2e-6)
    assert np.array_equal(got[0][1], got[1][1])  # sigma: the same sums in the same order
    # the overlapped schedule sorts on the side stream: same kernel, same order
    e = make_engine(P, Q)
    e.adaptive_refresh_begin()
    e.adaptive_refresh_commit()
    assert np.array_equal(e.adaptive_snapshot()[0].cpu().numpy(), order_o)


@pytest.mark.parametrize("I,d,split", [(41141, 16, 0), (30001, 8, 0), (65535, 4, 0), (20481, 8, 0), (36865, 8, 0),
                                       (20108, 128, 0), (20108, 16, 2), (20108, 16, 3), (20480, 8, 4), (9999, 24, 2),
                                       (2048, 16, 2), (50000, 4, 3)])
@pytest.mark.parametrize("kind", ["normal", "few-ties", "ties", "spike", "skewed", "equal"])
def test_split_binned_sort_is_the_stable_descending_order(I, d, split, kind):
    """k_sort_binned_split — G workgroups per column, each ordering a stretch of ranks: columns past one workgroup's
    LDS (20,480 < I <= 65,535: MSD's 41,141 items); `binned_split` forces G on smaller tables.  Same order as the
    oracle's stable descending argsort and as the radix path, same sigma as the one-workgroup kernel; heavily tied /
    all-equal columns take the per-column fallback (k_sort_flagged up to 36,864 items, the filtered split radix sort +
    merge beyond)."""
    rng = np.random.default_rng(I * 11 + d + len(kind) + split)
    Q = _binned_tables(I, d, kind, rng)
    P = np.zeros((4, d), np.float32)
    QT, sigma_o = oracle.adaptive_stats(Q)
    order_o = oracle.adaptive_order(QT)
    sig = {}
    for name, tune in (("split", {"binned_split": split}), ("radix", {"binned_sort": 0}),
                       ("one", {"binned_split": 1})):
        e = make_engine(P, Q)
        for k, v in tune.items():
            e.set_tuning(k, v)
        e.adaptive_refresh()
        order, sigma = e.adaptive_snapshot()
        assert np.array_equal(order.cpu().numpy(), order_o), (name, kind)
        sig[name] = sigma.cpu().numpy()
        if np.all(sigma_o > 0):
            assert close(sig[name], sigma_o, 2e-6)
    if I <= 20480:
        assert np.array_equal(sig["split"], sig["one"])  # the same sums in the same order


def test_atomics_lose_nothing_under_chip_wide_contention():
    """100k triples hammering 50 item rows from every CU / XCD at once: the accumulated gradients
    must be the exact sums (device-scope fp32 atomics resolve below the per-XCD L2s)."""
    U, I, d, n = 20000, 60, 128, 100_000
    rng = np.random.default_rng(0)
    P = ((rng.random((U, d)) - 0.5) * 0.2).astype(np.float32)
    Q = ((rng.random((I, d)) - 0.5) * 0.2).astype(np.float32)
    P[0] = 0
    Q[0] = 0
    users = rng.integers(1, U, size=n).astype(np.int32)
    pos = rng.integers(1, 51, size=n).astype(np.int32)
    neg = rng.integers(1, 51, size=n).astype(np.int32)
    reg = (0.01, 0.02, 0.03)
    e = make_engine(P, Q, None, reg)
    e.forward_grad(dev(users), dev(pos), dev(neg))
    gP, gQ, _ = (t.cpu().numpy() for t in e.get_grad())
    gPo, gQo, _ = oracle.dense_grad(P, Q, None, users, pos, neg, reg)
    scale = np.abs(gQo).max()
    assert scale > 10  # thousands of contributions per row
    assert np.abs(gQ - gQo).max() < 2e-4 * scale, np.abs(gQ - gQo).max() / scale
    assert close(gP, gPo, 1e-5)
    # same through the STREAM kernel.  Item rows start at zero (no fp32 absorption of small adds)
    # and lr is small, so every triple contributes lr * (gradient at the initial point) to first order.
    Q0 = np.zeros_like(Q)
    e2 = make_engine(P, Q0, None, (0.0, 0.0, 0.0))
    lr = 1e-4
    e2.set_optimizer(kind=0, lr=lr)
    e2.train_stream(dev(users), dev(pos), sampler=0, neg=dev(neg))
    _, gQ0, _ = oracle.dense_grad(P, Q0, None, users, pos, neg, (0.0, 0.0, 0.0))
    dQ = e2.Q.cpu().numpy().astype(np.float64) / -lr
    s0 = np.abs(gQ0).max()
    assert s0 > 3 and np.abs(dQ - gQ0).max() < 0.005 * s0, np.abs(dQ - gQ0).max() / s0
    # and with the hot-row delta rows in the path (bpr_plan_epoch measures popularity; the 16 most
    # popular of the 50 rows take their updates in delta rows that are folded after the launch)
    for hot_rows, replicas in ((16, 1), (50, 4)):
        e3 = make_engine(P, Q0, None, (0.0, 0.0, 0.0))
        e3.set_optimizer(kind=0, lr=lr)
        e3.set_hot_rows(hot_rows, replicas)
        e3.set_stream_opts(True, 8)
        pu, pi = e3.plan_epoch(dev(users), dev(pos), n, seed=3)
        order_back = {(int(a), int(b)): k for k, (a, b) in enumerate(zip(users, pos))}
        ng = np.asarray([neg[order_back[(int(a), int(b))]] for a, b in
                         zip(pu.cpu().numpy(), pi.cpu().numpy())], np.int32)
        e3.train_stream(pu, pi, sampler=0, neg=dev(ng))
        _, g3, _ = oracle.dense_grad(P, Q0, None, pu.cpu().numpy(), pi.cpu().numpy(), ng,
                                     (0.0, 0.0, 0.0))
        dQ3 = e3.Q.cpu().numpy().astype(np.float64) / -lr
        assert np.abs(dQ3 - g3).max() < 0.005 * s0, (hot_rows, np.abs(dQ3 - g3).max() / s0)


@pytest.mark.parametrize("I,d", [(20109, 128), (5001, 32), (12937, 64), (20480, 8), (300, 16)])
def test_refresh_follows_a_moving_table(I, d):
    """Refresh over a sequence of table states:
    small drifts, a large jump, an all-equal table (order = item ids),
    and back — every snapshot must equal the oracle's stable descending order exactly."""
    rng = np.random.default_rng(I * 31 + d)
    Q = (rng.standard_normal((I, d)) * 0.05).astype(np.float32)
    Q[0] = 0
    P = np.zeros((4, d), np.float32)
    e = make_engine(P, Q)

    def check():
        e.adaptive_refresh()
        order, sigma = e.adaptive_snapshot()
        Qh = e.Q.cpu().numpy()
        QT, sigma_o = oracle.adaptive_stats(Qh)
        assert np.array_equal(order.cpu().numpy(), oracle.adaptive_order(QT))
        if np.all(sigma_o > 0):
            assert close(sigma.cpu().numpy(), sigma_o, 2e-6)

    check()  # first refresh: linear splitters
    for step in range(3):  # small drift, ties included
        e.Q.add_(torch.randn(I, d, device="cuda") * 0.002)
        e.Q[7] = e.Q[5]
        e.Q[0] = 0
        check()
    e.Q.mul_(-3.0)  # every order reversed
    check()
    e.Q.add_(torch.randn(I, d, device="cuda") * 0.5)  # distribution shift
    check()
    e.Q.zero_()  # all keys equal → one bucket → radix fallback → ids ascending
    check()
    e.Q.copy_(torch.from_numpy(Q).cuda())
    check()


# ---- run boundaries against user boundaries (k_stream) ----------------------------------------------
def _tail_problem(d, seed, run_len):
    """A user-grouped chunk whose users have every length from 1 to 3 run lengths, so run
    boundaries cut users at every offset (a user inside one run: plain store; cut: atomic delta)."""
    rng = np.random.default_rng(seed)
    lens = np.concatenate([np.arange(1, 3 * run_len + 2), rng.integers(1, 3 * run_len, 120)])
    rng.shuffle(lens)
    U, I = len(lens) + 1, 400
    users = np.repeat(np.arange(1, U), lens).astype(np.int32)
    pos = np.concatenate([rng.choice(np.arange(1, I), size=k, replace=False) for k in lens]).astype(np.int32)
    indptr = np.zeros(U + 1, np.int64)
    indptr[2:] = np.cumsum(lens)
    order = np.lexsort((pos, users))
    indices = pos[order].astype(np.int32)
    P = ((rng.random((U, d)) - 0.5) * 0.5).astype(np.float32)
    Q = ((rng.random((I, d)) - 0.5) * 0.5).astype(np.float32)
    P[0] = 0
    Q[0] = 0
    return P, Q, indptr, indices, users, pos


@pytest.mark.parametrize("d,run_len", [(64, 8), (128, 8), (128, 4), (32, 24), (256, 8), (256, 5), (100, 7)])
def test_stream_runs_sequential_equals_b1_sgd(d, run_len):
    """One group walks the grouped chunk (max_inflight = 1): whatever the run length, every triple
    is processed exactly once and in stream order — the oracle's B = 1 stream — whether a user
    lies inside a run (plain store) or is cut by a run boundary (atomic delta)."""
    P, Q, indptr, indices, users, pos = _tail_problem(d, 7 * d + run_len, run_len)
    n = len(users)
    reg = (0.01, 0.02, 0.03)
    e = make_engine(P, Q, None, reg)
    e.bind_seen_csr(dev(indptr), dev(indices))
    e.set_optimizer(kind=0, lr=0.05)
    e.set_stream_opts(True, run_len)
    negs = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    sc = torch.zeros(4, device="cuda")
    e.train_stream(dev(users), dev(pos), sampler=1, neg=negs, seed=11, max_inflight=1, scalars=sc)
    Po, Qo = P.copy(), Q.copy()
    neg_o = np.zeros(n, np.int32)
    sco = oracle.train_stream_seq(Po, Qo, None, users, pos, neg_o, 1, 0.05, reg, indptr=indptr,
                                  indices=indices, seed=11)
    assert np.array_equal(negs.cpu().numpy(), neg_o)
    assert int(round(float(sc[3]))) == n
    assert close(e.P.cpu().numpy(), Po, 1e-5), maxerr(e.P.cpu().numpy(), Po)
    assert close(e.Q.cpu().numpy(), Qo, 1e-5), maxerr(e.Q.cpu().numpy(), Qo)
    assert close(sc.cpu().numpy()[:3], sco[:3], 1e-4)


@pytest.mark.parametrize("d,run_len", [(128, 8), (64, 4), (256, 8), (32, 24), (128, 1)])
def test_stream_runs_partition_the_chunk_at_full_concurrency(d, run_len):
    """Full chip, given negatives, lr -> 0 limit.  Every triple is walked exactly once whoever
    owns its user: every row moves by exactly its first-order update — P and Q equal the oracle's
    summed-gradient step up to O(lr^2), row by row — and the kernel counts n triples."""
    P, Q, indptr, indices, users, pos = _tail_problem(d, 3 * d + run_len, run_len)
    n = len(users)
    rng = np.random.default_rng(1)
    neg = rng.integers(1, Q.shape[0], n).astype(np.int32)
    lr = 1e-4
    e = make_engine(P, Q, None, (0.0, 0.0, 0.0))
    e.set_optimizer(kind=0, lr=lr)
    e.set_stream_opts(True, run_len)
    sc = torch.zeros(4, device="cuda")
    e.train_stream(dev(users), dev(pos), sampler=0, neg=dev(neg), scalars=sc)
    assert int(round(float(sc[3]))) == n
    Po, Qo = P.copy(), Q.copy()
    oracle.step_sgd_sparse(Po, Qo, None, users, pos, neg, lr, (0.0, 0.0, 0.0))  # one summed step
    # row by row: a user with a single triple must not drown next to one with seventy
    for got, want, start in ((e.P.cpu().numpy(), Po, P), (e.Q.cpu().numpy(), Qo, Q)):
        moved = np.abs(want - start).max(axis=1)
        err = np.abs(got - want).max(axis=1)
        assert moved.max() > 0
        assert np.all(err <= 0.1 * moved + 1e-7), (float((err / (moved + 1e-12)).max()))


# ---- split refresh (bpr_adaptive_refresh_begin / _commit) -------------------------------------------
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("I,d", [(20109, 128), (3001, 32), (45000, 16)])
def test_split_refresh_snapshots_the_table_at_begin(I, d, masked):
    """begin cuts the keys in stream order, the sort runs on the side stream (optionally on a
    CU-masked one), commit swaps: the samplers read the OLD snapshot until commit and, after it,
    exactly the order of the table as it was at begin — whatever happened to the table since."""
    from revisit_bpr import engine as eng

    rng = np.random.default_rng(I + d)
    Q0 = rng.normal(0, 0.1, (I, d)).astype(np.float32)
    Q0[0] = 0
    P = np.zeros((8, d), np.float32)
    e = make_engine(P, Q0)
    if masked:
        side = eng.MaskedStream(e.device, eng.cu_mask(0, 64))
        e.set_side_stream(side)
    e.adaptive_refresh()
    order0, sigma0 = (t.clone() for t in e.adaptive_snapshot())
    Q1 = (Q0 + rng.normal(0, 0.05, (I, d))).astype(np.float32)
    Q1[0] = 0
    e.Q.copy_(torch.from_numpy(Q1).cuda())
    assert not e.refresh_pending()
    e.adaptive_refresh_begin()
    assert e.refresh_pending()
    e.Q.mul_(-3.0)  # the table moves on while the sort runs
    o_mid, s_mid = e.adaptive_snapshot()
    assert torch.equal(o_mid, order0) and torch.equal(s_mid, sigma0)
    with pytest.raises(Exception):
        e.adaptive_refresh_begin()  # one split refresh at a time
    e.adaptive_refresh_commit()
    assert not e.refresh_pending()
    order1, sigma1 = e.adaptive_snapshot()
    QT, sig = oracle.adaptive_stats(Q1)
    assert np.array_equal(order1.cpu().numpy(), oracle.adaptive_order(QT))
    assert close(sigma1.cpu().numpy(), sig, 1e-5)
    with pytest.raises(Exception):
        e.adaptive_refresh_commit()  # nothing pending
    e.adaptive_refresh()  # the synchronous call still works and sees the table as it is now
    QT2, _ = oracle.adaptive_stats(e.Q.cpu().numpy())
    assert np.array_equal(e.adaptive_snapshot()[0].cpu().numpy(), oracle.adaptive_order(QT2))


def test_synchronous_refresh_after_a_pending_split_refresh():
    """A lagged schedule ends every epoch with a split refresh in flight: the next
    synchronous bpr_adaptive_refresh — StrictTrainer, bpr_train_strict's refresh_every — commits it
    implicitly and then snapshots the table as it is now; nothing is left pending."""
    I, d = 5000, 64
    rng = np.random.default_rng(5)
    Q0 = rng.normal(0, 0.1, (I, d)).astype(np.float32)
    Q0[0] = 0
    e = make_engine(np.zeros((8, d), np.float32), Q0)
    e.adaptive_refresh()
    e.adaptive_refresh_begin()
    assert e.refresh_pending()
    Q1 = (Q0 * -2.0 + 0.01).astype(np.float32)
    Q1[0] = 0
    e.Q.copy_(torch.from_numpy(Q1).cuda())
    e.adaptive_refresh()  # was: BPR_ERR_INVALID ("a split or sharded refresh is pending")
    assert not e.refresh_pending()
    QT, _ = oracle.adaptive_stats(Q1)
    assert np.array_equal(e.adaptive_snapshot()[0].cpu().numpy(), oracle.adaptive_order(QT))
    # bpr_train_strict with refresh_every on a ctx that a lagged trainer left behind
    e.adaptive_refresh_begin()
    P, Q, indptr, indices, users, pos, _ = rand_problem(8, I, d, 5, seed=3, B=64)
    e.bind_seen_csr(dev(indptr), dev(indices))
    e.set_optimizer(kind=0, lr=0.01)
    e.train_strict(dev(users), dev(pos), 16, sampler=2, adaptive_p=0.1, seed=1, refresh_every=2)
    assert not e.refresh_pending()
    torch.cuda.synchronize()


@pytest.mark.parametrize("I,d,parts", [(20109, 128, 2), (3001, 32, 4), (45000, 16, 2), (20109, 128, 8)])
def test_sharded_refresh_parts_equal_the_full_sort(I, d, parts):
    """bpr_adaptive_refresh_part sorts a slice of the factors into the back snapshot; all slices
    together, published, are bit for bit the snapshot of the full refresh (multi-GPU: every rank
    sorts one slice of its replica and an all-gather fills in the others)."""
    rng = np.random.default_rng(I + d)
    Q = rng.normal(0, 0.1, (I, d)).astype(np.float32)
    Q[0] = 0
    e = make_engine(np.zeros((8, d), np.float32), Q)
    e.adaptive_refresh()
    want_o, want_s = (t.clone() for t in e.adaptive_snapshot())
    e.Q.mul_(-1.5)  # the front snapshot now differs from what a refresh would give
    per = d // parts
    back_o, back_s = e._snapshot_views(back=True)
    gathered_o, gathered_s = torch.empty_like(back_o), torch.empty_like(back_s)
    from revisit_bpr import native

    for r in range(parts):  # one "rank" after the other on this GPU: each call cuts + sorts a slice
        native.check(e._lib.bpr_adaptive_refresh_part(e._ctx, r * per, (r + 1) * per))
        gathered_o[r * per:(r + 1) * per] = back_o[r * per:(r + 1) * per]
        gathered_s[r * per:(r + 1) * per] = back_s[r * per:(r + 1) * per]
        if r < parts - 1:  # (a real rank publishes after the gather; here: drop the pending flag)
            native.check(e._lib.bpr_adaptive_refresh_publish(e._ctx))
            back_o, back_s = e._snapshot_views(back=True)
    back_o.copy_(gathered_o)
    back_s.copy_(gathered_s)
    native.check(e._lib.bpr_adaptive_refresh_publish(e._ctx))
    got_o, got_s = e.adaptive_snapshot()
    QT, sig = oracle.adaptive_stats(e.Q.cpu().numpy())
    assert np.array_equal(got_o.cpu().numpy(), oracle.adaptive_order(QT))
    assert close(got_s.cpu().numpy(), sig, 1e-5)
    assert not torch.equal(got_o, want_o)


@pytest.mark.parametrize("lag,fused", [(1.0, False), (1.0, True), (0.4, False)])
def test_split_refresh_pipeline_sequential_equals_oracle(lag, fused):
    """The StreamTrainer's lagged schedule, one group at a time (max_inflight = 1), against the
    oracle walking the same triples with the snapshot the schedule prescribes: cut at `begin`,
    in force from `commit`."""
    d, U, I, n, chunk = 256, 50, 120, 600, 200
    P, Q, indptr, indices, users, pos, _ = rand_problem(U, I, d, 25, seed=11, B=n)
    P *= 8
    Q *= 8
    reg = (0.01, 0.02, 0.03)
    e = make_engine(P, Q, None, reg)
    e.bind_seen_csr(dev(indptr), dev(indices))
    e.set_optimizer(kind=0, lr=0.05)
    negs = torch.zeros(n, dtype=torch.int32, device="cuda")
    u_d, p_d = dev(users), dev(pos)
    Po, Qo = P.copy(), Q.copy()
    neg_o = np.zeros(n, np.int32)

    def snap():
        QT, sigma = oracle.adaptive_stats(Qo)
        return sigma, oracle.adaptive_order(QT)

    def both(lo, hi, cur):
        e.train_stream(u_d[lo:hi], p_d[lo:hi], sampler=2, neg=negs[lo:hi], adaptive_p=0.1, seed=9,
                       offset=lo, max_inflight=1, cut=fused)
        oracle.train_stream_seq(Po, Qo, None, users[lo:hi], pos[lo:hi], neg_o[lo:hi], 2, 0.05, reg,
                                adaptive_p=0.1, sigma=cur[0], order=cur[1], indptr=indptr,
                                indices=indices, seed=9, offset=lo)

    cur = pend = None
    for lo in range(0, n, chunk):
        hi = lo + chunk
        if pend is None:
            e.adaptive_refresh()
            cur = snap()
        else:
            e.adaptive_refresh_commit()
            cur = pend
        cut = lo if lag >= 1.0 else lo + int(round((1.0 - lag) * chunk))
        if cut > lo:
            both(lo, cut, cur)
        e.adaptive_refresh_begin()
        pend = snap()
        both(cut, hi, cur)
    got = negs.cpu().numpy()
    assert (got == neg_o).mean() > 0.97
    if np.array_equal(got, neg_o):
        assert close(e.P.cpu().numpy(), Po, 2e-5), maxerr(e.P.cpu().numpy(), Po)
        assert close(e.Q.cpu().numpy(), Qo, 2e-5)


@pytest.mark.parametrize("d,sampler,n", [(128, 2, 120_000), (128, 1, 30_000), (256, 2, 40_000),
                                          (100, 0, 25_000), (32, 2, 900)])
def test_stream_cut_at_full_concurrency_sees_the_final_table(d, sampler, n):
    """bpr_train_stream_cut with the chip full: the keys of the next snapshot are cut by the launch's
    epilogue AFTER every workgroup's atomics — the snapshot committed afterwards is bit for bit the
    oracle's order of the item table as the launch left it (hot rows folded), launch after launch,
    and the loss statistics count every triple.  (r4 also built the cut as an in-kernel tail of the
    launch — last-workgroups-done, sharded tickets — and held it to this test; it was slower than the
    separate epilogue kernel and is not in the tree: profiles/r04_tailcut.md.)"""
    rng = np.random.default_rng(d + n)
    U, I = 6000, 3000
    P = rng.normal(0, 0.1, (U, d)).astype(np.float32)
    Q = rng.normal(0, 0.1, (I, d)).astype(np.float32)
    P[0] = 0
    Q[0] = 0
    lens = rng.integers(1, 40, U)
    lens[0] = 0
    rows = [np.sort(rng.choice(np.arange(1, I), size=int(k), replace=False)).astype(np.int32) for k in lens]
    indptr = np.zeros(U + 1, np.int64)
    indptr[1:] = np.cumsum(lens)
    indices = np.concatenate(rows)
    users = rng.integers(1, U, n).astype(np.int32)
    pos = (1 + (rng.zipf(1.3, n) % (I - 1))).astype(np.int32)  # skewed: the hot block is in use
    e = make_engine(P, Q, None, (0.01, 0.02, 0.03))
    e.bind_seen_csr(dev(indptr), dev(indices))
    e.set_optimizer(kind=0, lr=0.05)
    e.set_stream_opts(True, 0)
    pu, pi = e.plan_epoch(dev(users), dev(pos), n, seed=3)  # grouped by user, hot block built
    neg = dev(rng.integers(1, I, n).astype(np.int32)) if sampler == 0 else None
    sc = torch.zeros(4, device="cuda")
    e.adaptive_refresh()
    for launch in range(3):
        e.train_stream(pu, pi, sampler=sampler, neg=neg, adaptive_p=0.05, seed=5, offset=launch * n,
                       scalars=sc, cut=True)
        Qnow = e.Q.cpu().numpy()  # (stream order: after the launch and its fold)
        e.adaptive_refresh_begin()  # only queues the sort: the keys were cut by the launch
        e.Q.mul_(1.0)
        e.adaptive_refresh_commit()
        QT, sig = oracle.adaptive_stats(Qnow)
        got_o, got_s = e.adaptive_snapshot()
        assert np.array_equal(got_o.cpu().numpy(), oracle.adaptive_order(QT)), launch
        assert close(got_s.cpu().numpy(), sig, 1e-5)
        assert int(sc[3]) == (launch + 1) * n
    assert torch.isfinite(e.Q).all() and torch.isfinite(sc).all()


@pytest.mark.parametrize("fold", [1, 0])
def test_async_cut_reads_the_table_whole_and_folds_on_demand(fold):
    """(fold = 1, r6 default: only the transpose of the keys goes to the side stream, the hot block is folded on the
    launch stream after every launch — the same contract from outside; fold = 0: r4's form, below.)
    bpr_train_stream_acut: the cut is a read-only pass on the side stream (keys = Q + hot deltas) and
    the launch folds nothing.  With nothing running beside it the snapshot it yields is exactly the
    oracle's order of the table the launch left; any other entry point (here: the synchronous
    refresh, the item table read through hot_fold) sees the table whole; the loss statistics arrive
    through the side stream; and a run of asynchronous launches moves the table like a run of
    synchronous ones (same negatives given: the same sums up to fp32 association)."""
    rng = np.random.default_rng(8)
    U, I, d, n = 4000, 2500, 128, 60_000
    P = rng.normal(0, 0.1, (U, d)).astype(np.float32)
    Q = rng.normal(0, 0.1, (I, d)).astype(np.float32)
    P[0] = 0
    Q[0] = 0
    users = rng.integers(1, U, n).astype(np.int32)
    pos = (1 + (rng.zipf(1.3, n) % (I - 1))).astype(np.int32)
    neg = rng.integers(1, I, n).astype(np.int32)
    indptr = np.zeros(U + 1, np.int64)
    res = []
    for mode in ("async", True):
        e = make_engine(P, Q, None, (0.01, 0.02, 0.03))
        e.bind_seen_csr(dev(indptr), dev(np.zeros(0, np.int32)))
        e.set_optimizer(kind=0, lr=0.01)
        e.set_stream_opts(True, 0)
        e.set_tuning("acut_fold", fold)
        pu, pi = e.plan_epoch(dev(users), dev(pos), n, seed=3)  # builds the hot block
        sc = torch.zeros(4, device="cuda")
        e.adaptive_refresh()
        for k in range(3):
            e.train_stream(pu, pi, sampler=0, neg=dev(neg), scalars=sc, cut=mode, max_inflight=1)
            torch.cuda.synchronize()  # nothing beside the cut: the snapshot is exact
            e.adaptive_refresh_begin()
            e.adaptive_refresh_commit()
            order = e.adaptive_snapshot()[0].cpu().numpy()   # (an entry point: folds what is pending)
            e.hot_fold()
            torch.cuda.synchronize()
            QT, _ = oracle.adaptive_stats(e.Q.cpu().numpy())
            assert np.array_equal(order, oracle.adaptive_order(QT)), (mode, k)
        assert int(sc[3]) == 3 * n
        res.append((e.P.cpu().numpy(), e.Q.cpu().numpy(), sc.cpu().numpy()))
    assert close(res[0][0], res[1][0], 1e-6) and close(res[0][1], res[1][1], 1e-6)
    assert close(res[0][2], res[1][2], 1e-4)


@pytest.mark.parametrize("hot_rows", [0, 256])
def test_sync_refresh_right_behind_an_async_cut_sorts_finished_keys(hot_rows):
    """after bpr_train_stream_acut the keys of the next snapshot are cut on the SIDE stream.
    A synchronous refresh (bpr_adaptive_refresh: StrictTrainer, bpr_train_strict's refresh_every) sorts
    them on the launch stream — it must wait for that cut, with or without a hot block to fold, and a
    launch outside the acut pipeline must not reuse the partials the cut still sums.  No host
    synchronisation between the calls here: only the library's own events order them."""
    rng = np.random.default_rng(18)
    U, I, d, n = 6000, 9000, 128, 200_000
    P = rng.normal(0, 0.1, (U, d)).astype(np.float32)
    Q = rng.normal(0, 0.1, (I, d)).astype(np.float32)
    P[0] = 0
    Q[0] = 0
    users = rng.integers(1, U, n).astype(np.int32)
    pos = (1 + (rng.zipf(1.3, n) % (I - 1))).astype(np.int32)
    neg = rng.integers(1, I, n).astype(np.int32)
    e = make_engine(P, Q, None, (0.01, 0.02, 0.03))
    e.bind_seen_csr(dev(np.zeros(U + 1, np.int64)), dev(np.zeros(0, np.int32)))
    e.set_optimizer(kind=0, lr=0.01)
    e.set_stream_opts(True, 0)
    e.set_hot_rows(hot_rows, 1)
    pu, pi = e.plan_epoch(dev(users), dev(pos), n, seed=3)
    sc = torch.zeros(4, device="cuda")
    e.adaptive_refresh()
    torch.cuda.synchronize()
    for k in range(4):
        e.train_stream(pu, pi, sampler=0, neg=dev(neg), scalars=sc, cut="async")
        e.adaptive_refresh()  # synchronous, on the launch stream, right behind the asynchronous cut
        order = e.adaptive_snapshot()[0].cpu().numpy()
        e.hot_fold()
        torch.cuda.synchronize()
        QT, _ = oracle.adaptive_stats(e.Q.cpu().numpy())
        assert np.array_equal(order, oracle.adaptive_order(QT)), k
    # ... and a plain launch behind an asynchronous one: both sets of statistics arrive whole
    e.train_stream(pu, pi, sampler=0, neg=dev(neg), scalars=sc, cut="async")
    e.train_stream(pu, pi, sampler=0, neg=dev(neg), scalars=sc)
    torch.cuda.synchronize()
    assert int(sc[3]) == 6 * n


@pytest.mark.parametrize("cut", [False, True])
def test_item_bias_write_back_rides_on_the_epilogue_and_tracking_skips_only_clean_refills(cut):
    """r5: k_stream's one-item-per-line copy of the item_bias is written back by the launch's own
    epilogue (k_stream_epilogue / _cut) instead of a kernel of its own, and with bias tracking on the
    refill before a launch is skipped while the copy is current.  The sequence launch, launch, [torch
    writes the vector + bias_written], launch must leave exactly what the same sequence leaves with a
    refill before every launch (sequential mode: one wave walks the stream, so both are deterministic),
    and both must follow the oracle's sequential SGD."""
    rng = np.random.default_rng(31)
    U, I, d, n = 300, 220, 64, 3000
    P = rng.normal(0, 0.2, (U, d)).astype(np.float32)
    Q = rng.normal(0, 0.2, (I, d)).astype(np.float32)
    b = rng.normal(0, 0.1, I).astype(np.float32)
    P[0] = 0
    Q[0] = 0
    users = np.sort(rng.integers(1, U, n)).astype(np.int32)
    pos = rng.integers(1, I, n).astype(np.int32)
    neg = rng.integers(1, I, n).astype(np.int32)
    bump = rng.normal(0, 0.05, I).astype(np.float32)
    reg, lr = (0.01, 0.02, 0.03), 0.05
    out = []
    for track in (False, True):
        e = make_engine(P, Q, b, reg)
        e.bind_seen_csr(dev(np.zeros(U + 1, np.int64)), dev(np.zeros(0, np.int32)))
        e.set_optimizer(kind=0, lr=lr)
        e.set_stream_opts(True, 0)
        e.set_hot_rows(0, 1)
        e.set_bias_tracking(track)
        sc = torch.zeros(4, device="cuda")
        if cut:
            e.adaptive_refresh()
        for k in range(3):
            if k == 2:
                e.item_bias.add_(dev(bump))  # a write the library cannot see ...
                e.bias_written()             # ... declared
            e.train_stream(dev(users), dev(pos), sampler=0, neg=dev(neg), scalars=sc, max_inflight=1, cut=cut)
        torch.cuda.synchronize()
        out.append((e.P.cpu().numpy(), e.Q.cpu().numpy(), e.item_bias.cpu().numpy(), sc.cpu().numpy()))
    for a, c in zip(out[0], out[1]):
        assert np.array_equal(a, c)
    Po, Qo, bo = P.copy(), Q.copy(), b.copy()
    for k in range(3):
        if k == 2:
            bo += bump
        oracle.train_stream_seq(Po, Qo, bo, users, pos, neg, oracle.NEG_GIVEN, lr, reg)
    assert close(out[1][0], Po, 2e-5) and close(out[1][1], Qo, 2e-5) and close(out[1][2], bo, 2e-5)


# ---- heavy users: precomputed seen bitmaps in HBM -------------------------------------------------
@pytest.mark.parametrize("seen,heavy_t", [("", None), ("list", None), ("", "-1"), ("", "40"), ("list", "600")])
@pytest.mark.parametrize("d", [64, 256])
def test_stream_heavy_users_pick_like_the_oracle(d, seen, heavy_t, monkeypatch):
    """Users with more than 256 seen items read "seen?" from their precomputed bitmap in HBM, the
    others from the LDS structure built at the user change: a chunk that mixes both (and switches
    between them inside a run) must draw exactly the oracle's uniform negatives and — up to fp32
    bin-edge flips — its adaptive ones, whatever the threshold."""
    if seen:
        monkeypatch.setenv("BPR_SEEN", seen)
    if heavy_t is not None:
        monkeypatch.setenv("BPR_HEAVY_T", heavy_t)
    rng = np.random.default_rng(d)
    U, I, n = 300, 2500, 12000
    lens = np.where(rng.random(U) < 0.3, rng.integers(257, 1500, U), rng.integers(0, 120, U))
    lens[0] = 0
    rows = [np.sort(rng.choice(np.arange(1, I), size=int(k), replace=False)).astype(np.int32) for k in lens]
    indptr = np.zeros(U + 1, np.int64)
    indptr[1:] = np.cumsum(lens)
    indices = np.concatenate(rows)
    P = (rng.normal(0, 0.3, (U, d))).astype(np.float32)
    Q = (rng.normal(0, 0.3, (I, d))).astype(np.float32)
    P[0] = 0
    Q[0] = 0
    users = np.sort(rng.integers(1, U, n)).astype(np.int32)  # grouped by user
    pos = rng.integers(1, I, n).astype(np.int32)
    e = make_engine(P, Q, None, (0.01, 0.01, 0.01))
    e.bind_seen_csr(dev(indptr), dev(indices))
    e.set_optimizer(kind=0, lr=0.0)
    e.set_stream_opts(True, 8)
    e.adaptive_refresh()
    QT, sigma = oracle.adaptive_stats(Q)
    order = oracle.adaptive_order(QT)
    for sampler in (1, 2):
        negs = torch.zeros(n, dtype=torch.int32, device="cuda")
        e.train_stream(dev(users), dev(pos), sampler=sampler, neg=negs, adaptive_p=0.02, seed=5, offset=7)
        got = negs.cpu().numpy()
        if sampler == 1:
            assert np.array_equal(got, oracle.sample_uniform(indptr, indices, I, users, seed=5, offset=7))
        else:
            want, _, _ = oracle.sample_adaptive(P, sigma, order, indptr, indices, users, 0.02, seed=5,
                                                offset=7)
            assert (got == want).mean() > 0.995, (got == want).mean()
        for t in range(0, n, 61):
            assert got[t] != 0 and got[t] not in indices[indptr[users[t]]:indptr[users[t] + 1]]
    # a second CSR bound to the same engine: the bitmaps are rebuilt, not reused
    indices2 = indices.copy()
    for u in range(1, U):
        k = int(lens[u])
        if k:
            indices2[indptr[u]:indptr[u + 1]] = np.sort(rng.choice(np.arange(1, I), size=k, replace=False))
    e.bind_seen_csr(dev(indptr), dev(indices2))
    negs = torch.zeros(n, dtype=torch.int32, device="cuda")
    e.train_stream(dev(users), dev(pos), sampler=1, neg=negs, seed=5, offset=7)
    assert np.array_equal(negs.cpu().numpy(),
                          oracle.sample_uniform(indptr, indices2, I, users, seed=5, offset=7))


# ---- partial adaptive snapshots (r5): exact ends + bucketed middle, the walk finishes inside a bin ----
@pytest.mark.parametrize("seen", ["", "list"])
@pytest.mark.parametrize("I,d,target,ties", [(2500, 128, 4, False), (2500, 128, 64, False), (9000, 64, 16, False),
                                             (20108, 128, 1024, False), (4000, 256, 8, False),
                                             (3000, 128, 32, True)])
def test_partial_snapshot_picks_exactly_what_the_full_snapshot_picks(I, d, target, ties, seen, monkeypatch):
    """bpr_adaptive_refresh_begin with bpr_set_tuning("partial_snapshot", 1) sorts only the two ends of
    every column and buckets the middle (k_sort_partial); k_stream's walk finishes inside a bin by
    ranking its keys on the fly.  Whatever the size of the exact ends — `target` 4 sends nearly every walk
    through the finish — the negatives must be, triple for triple, those drawn from the fully sorted
    snapshot of the same table, i.e. the oracle's; heavy users (long walks, both ends) and light ones,
    walks from the top and from the bottom; a column with many equal keys is given up by the partial
    sort and sorted whole.  The snapshot the API hands out afterwards is complete (sorted on demand)."""
    if seen:
        monkeypatch.setenv("BPR_SEEN", seen)
    rng = np.random.default_rng(I + target)
    U, n = 400, 40_000
    lens = np.where(rng.random(U) < 0.25, rng.integers(I // 3, (2 * I) // 3, U), rng.integers(0, 90, U))
    lens[0] = 0
    rows = [np.sort(rng.choice(np.arange(1, I), size=int(k), replace=False)).astype(np.int32) for k in lens]
    indptr = np.zeros(U + 1, np.int64)
    indptr[1:] = np.cumsum(lens)
    indices = np.concatenate(rows)
    P = rng.normal(0, 0.3, (U, d)).astype(np.float32)
    Q = rng.normal(0, 0.3, (I, d)).astype(np.float32)
    if ties:
        Q[:, :7] = np.round(Q[:, :7] * 3) / 3   # a few dozen distinct values per column: bins overflow
        Q[:, 7] = 0.25                           # an all-equal column
    P[0] = 0
    Q[0] = 0
    users = np.sort(rng.integers(1, U, n)).astype(np.int32)
    pos = rng.integers(1, I, n).astype(np.int32)
    out = {}
    for partial in (0, 1):
        e = make_engine(P, Q, None, (0.01, 0.01, 0.01))
        e.bind_seen_csr(dev(indptr), dev(indices))
        e.set_optimizer(kind=0, lr=0.0)
        e.set_stream_opts(True, 0)
        e.set_tuning("partial_snapshot", partial)
        e.set_tuning("partial_target", target)
        e.adaptive_refresh()
        e.adaptive_refresh_begin()
        e.adaptive_refresh_commit()
        assert e.snapshot_partial() == bool(partial)
        negs = torch.zeros(n, dtype=torch.int32, device="cuda")
        e.train_stream(dev(users), dev(pos), sampler=2, neg=negs, adaptive_p=0.02, seed=5, offset=7)
        out[partial] = negs.cpu().numpy()
        if partial:
            assert e.snapshot_partial()  # the launch read it as it was
            order = e.adaptive_snapshot()[0].cpu().numpy()  # completed on demand
            assert not e.snapshot_partial()
            QT, _ = oracle.adaptive_stats(Q)
            assert np.array_equal(order, oracle.adaptive_order(QT))
            # ... and the sampler kernel behind the Python API reads the completed snapshot
            got2 = e.sample_adaptive(dev(users[:2048]), 0.02, seed=5, offset=7).cpu().numpy()
            assert np.array_equal(got2, out[0][:2048])
    assert np.array_equal(out[0], out[1]), (out[0] != out[1]).mean()
    QT, sigma = oracle.adaptive_stats(Q)
    want, _, _ = oracle.sample_adaptive(P, sigma, oracle.adaptive_order(QT), indptr, indices, users, 0.02, seed=5,
                                        offset=7)
    assert (out[1] == want).mean() > 0.995


# ---- uniform sampler: exact pick when rejection cannot succeed ------------------------------------
@pytest.mark.parametrize("seen", ["", "csr", "list"])
@pytest.mark.parametrize("d", [32, 256])
def test_uniform_sampler_user_who_has_seen_nearly_everything(d, seen, monkeypatch):
    """UniformSampler.sample (neg_samplers.py:31-37) always returns an unseen item, however few
    are left.  A user with 3 unseen items out of 60,000 defeats 4,096 rejection rounds almost
    surely (0.815 probability); the exact rank pick takes over and never returns the pad item."""
    if seen:
        monkeypatch.setenv("BPR_SEEN", seen)
    I, U = 60_001, 6
    rng = np.random.default_rng(d)
    unseen = {1: [5, 30_000, 60_000], 2: [1, 2, 3], 3: [59_998, 59_999, 60_000], 4: [777]}
    rows = []
    for u in range(U):
        if u in unseen:
            rows.append(np.setdiff1d(np.arange(1, I), unseen[u]).astype(np.int32))
        elif u == 5:
            rows.append(np.arange(1, I, dtype=np.int32))  # nothing left: item 0, as documented
        else:
            rows.append(np.zeros(0, np.int32))
    indptr = np.zeros(U + 1, np.int64)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    indices = np.concatenate(rows)
    P = rng.normal(0, 0.1, (U, d)).astype(np.float32)
    Q = rng.normal(0, 0.1, (I, d)).astype(np.float32)
    e = make_engine(P, Q)
    e.bind_seen_csr(dev(indptr), dev(indices))
    users = np.tile(np.arange(1, U, dtype=np.int32), 40)
    got = e.sample_uniform(dev(users), seed=3, offset=50).cpu().numpy()
    want = oracle.sample_uniform(indptr, indices, I, users, seed=3, offset=50)
    assert np.array_equal(got, want)
    for u, items in unseen.items():
        picks = got[users == u]
        assert set(picks.tolist()) <= set(items) and len(set(picks.tolist())) == len(items)
    assert (got[users == 5] == 0).all()
    # the same inside a STREAM launch (lr = 0: the tables stay put)
    e.set_optimizer(kind=0, lr=0.0)
    pos = np.array([unseen.get(int(u), [1])[0] for u in users], np.int32)
    keep = users != 5
    su, sp = np.sort(users[keep]), pos[keep][np.argsort(users[keep], kind="stable")]
    e.set_stream_opts(True, 8)
    negs = torch.zeros(len(su), dtype=torch.int32, device="cuda")
    e.train_stream(dev(su), dev(sp), sampler=1, neg=negs, seed=3, offset=50)
    assert np.array_equal(negs.cpu().numpy(),
                          oracle.sample_uniform(indptr, indices, I, su, seed=3, offset=50))
