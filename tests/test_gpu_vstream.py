"""BATCHED STREAM (bpr_train_stream_batched) against the CPU oracle — deterministic.

The kernel's sequential limit (max_inflight = 1: one group walks the stream) IS the reference's
mini-batch loop: every gradient of a virtual batch sees the pre-step rows, each row takes ONE dense
torch.optim step per batch with the summed gradient, untouched rows are moved by the lazy replay.
So it is held here to `oracle.step` (the dense restatement pinned to the reference's torch.optim
trajectories by tests/golden/) for every optimizer, with given / uniform / adaptive negatives, with
and without item bias, across chunk boundaries and mode switches.  At full concurrency the result
is exact whenever a launch holds one virtual batch (B >= n), which pins the concurrent protocol
(header CAS, double-buffered accumulators, close-and-apply) under heavy row contention.

Tolerances as tests/test_gpu_parity.py: 2e-5 relative to max(1, |w|) after a few hundred steps.
"""
import math

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from test_gpu_parity import close, close_mostly, dev, i32, make_engine, maxerr, rand_problem  # noqa: E402

OPTS = {
    "sgd": dict(kind=0, lr=0.05),
    "nesterov": dict(kind=1, lr=0.01, momentum=0.9, nesterov=True),
    "momentum_damp": dict(kind=1, lr=0.02, momentum=0.5, dampening=0.3),
    "adam_09": dict(kind=2, lr=0.003, betas=(0.9, 0.999)),
    "adam_01": dict(kind=2, lr=0.003, betas=(0.1, 0.999)),
    "adam_00": dict(kind=2, lr=0.003, betas=(0.0, 0.99)),
    "rmsprop": dict(kind=3, lr=0.0005, alpha=0.9),
    "rmsprop_mom": dict(kind=3, lr=0.0003, alpha=0.9, momentum=0.8),
}
REG = (0.0016, 0.0001, 0.00375)


def agree(got, want, cfg, tol=2e-5):
    """Exact-to-tolerance for SGD / momentum; Adam and RMSprop additionally allow the handful of
    ill-conditioned elements described in test_gpu_parity.close_mostly."""
    if cfg["kind"] in (2, 3):
        return close_mostly(got, want, tol, cap=10 * cfg["lr"])
    return close(got, want, tol)


def distinct_neg(pos, neg, I):
    """The samplers never return the positive itself (it is a seen item)."""
    return np.where(neg == pos, neg % (I - 1) + 1, neg).astype(np.int32)


def oracle_opt(cfg):
    return oracle.make_opt(cfg["kind"], **{k: v for k, v in cfg.items() if k != "kind"})


def oracle_state(P, Q, b):
    st = {k: np.zeros_like(P if k.endswith("P") else Q) for k in ("mP", "vP", "mQ", "vQ")}
    if b is not None:
        st["mb"], st["vb"] = np.zeros_like(b), np.zeros_like(b)
    return st


def oracle_batches(P, Q, b, users, pos, neg, B, cfg, t0=0, st=None):
    """The reference loop on the CPU: consecutive batches of B, dense optimizer."""
    opt = oracle_opt(cfg)
    st = st if st is not None else oracle_state(P, Q, b)
    loss = 0.0
    for k, lo in enumerate(range(0, len(users), B)):
        sl = slice(lo, lo + B)
        _, _, sc = oracle.step(P, Q, b, users[sl], pos[sl], neg[sl], opt, t0 + k + 1, st, REG)
        loss += sc[0]
    return st, loss


@pytest.mark.parametrize("opt_name", list(OPTS))
@pytest.mark.parametrize("d,bias", [(32, False), (50, True), (128, True), (256, False)])
def test_sequential_limit_is_the_reference_minibatch_loop(opt_name, d, bias):
    cfg = OPTS[opt_name]
    U, I, B, steps = 300, 200, 24, 60
    P, Q, _, _, _, _, _ = rand_problem(U, I, d, 10, seed=d + 3, B=8)
    P *= 4
    Q *= 4
    b = np.linspace(-0.1, 0.1, I).astype(np.float32) if bias else None
    rng = np.random.default_rng(5)
    n = steps * B - 7  # ragged last batch
    users = rng.integers(1, U, n).astype(np.int32)
    pos = rng.integers(1, I, n).astype(np.int32)
    neg = rng.integers(1, I, n).astype(np.int32)
    users[:6] = users[6]  # duplicates inside the first batch: users, positives, pos == another's neg
    pos[8:14] = pos[14]
    neg[16:20] = pos[14]
    users[40] = 0  # a pad user and a pad item: the embedding rows' gradients are dropped
    pos[41] = 0
    neg = distinct_neg(pos, neg, I)
    e = make_engine(P, Q, b, REG)
    e.set_optimizer(**cfg)
    e.alloc_opt_state()
    sc = torch.zeros(4, device="cuda")
    cut = 20 * B  # two launches: pending steps and the step counter carry across
    e.train_stream_batched(dev(users[:cut]), dev(pos[:cut]), B, sampler=0, neg=dev(neg[:cut]),
                           max_inflight=1, scalars=sc)
    e.train_stream_batched(dev(users[cut:]), dev(pos[cut:]), B, sampler=0, neg=dev(neg[cut:]),
                           max_inflight=1, scalars=sc)
    e.flush_lazy()
    Po, Qo, bo = P.copy(), Q.copy(), None if b is None else b.copy()
    _, loss = oracle_batches(Po, Qo, bo, users, pos, neg, B, cfg)
    assert e.step_count == steps
    Pg, Qg = e.P.cpu().numpy(), e.Q.cpu().numpy()
    assert np.abs(Po - P).max() > 1e-3
    assert agree(Pg, Po, cfg), maxerr(Pg, Po)
    assert agree(Qg, Qo, cfg), maxerr(Qg, Qo)
    if bias:
        assert agree(e.item_bias.cpu().numpy(), bo, cfg)
    assert not Pg[0].any() and not Qg[0].any()
    sc = sc.cpu().numpy()
    assert sc[3] == n and abs(sc[0] - loss) <= 2e-4 * loss


@pytest.mark.parametrize("opt_name", ["sgd", "adam_01", "nesterov"])
@pytest.mark.parametrize("sampler", [1, 2])
@pytest.mark.parametrize("seen", ["", "csr", "list"])
def test_sequential_limit_with_on_device_sampling(opt_name, sampler, seen, monkeypatch):
    """Uniform / adaptive negatives drawn in the kernel from the rows as of the previous step (the
    reference samples before the forward of the same batch), Philox counter = offset + position."""
    if seen:
        monkeypatch.setenv("BPR_SEEN", seen)
    from revisit_bpr.datasets import synthetic

    cfg = OPTS[opt_name]
    data = synthetic.generate(300, 200, 6000, median_per_user=12, seed=2)
    d, B, n = 64, 32, 1500
    rng = np.random.default_rng(1)
    P = ((rng.random((data.num_users, d)) - 0.5) / d * 4).astype(np.float32)
    Q = ((rng.random((data.num_items, d)) - 0.5) / d * 4).astype(np.float32)
    P[0] = 0
    Q[0] = 0
    perm = rng.permutation(data.nnz)[:n]
    users, pos = data.users[perm].copy(), data.items[perm].copy()
    e = make_engine(P, Q, None, REG)
    e.bind_seen_csr(dev(data.indptr), dev(data.indices))
    e.set_optimizer(**cfg)
    e.alloc_opt_state()
    e.adaptive_refresh()
    neg = torch.zeros(n, dtype=torch.int32, device="cuda")
    e.train_stream_batched(dev(users), dev(pos), B, sampler=sampler, neg=neg, adaptive_p=0.05,
                           seed=11, offset=1000, max_inflight=1)
    e.flush_lazy()
    Po, Qo = P.copy(), Q.copy()
    QT, sigma = oracle.adaptive_stats(Q)
    order = oracle.adaptive_order(QT)
    opt, st = oracle_opt(cfg), oracle_state(Po, Qo, None)
    neg_o = np.zeros(n, np.int32)
    for k, lo in enumerate(range(0, n, B)):
        sl = slice(lo, lo + B)
        if sampler == 1:
            nb = oracle.sample_uniform(data.indptr, data.indices, data.num_items, users[sl], 11,
                                       1000 + lo)
        else:
            nb, _, _ = oracle.sample_adaptive(Po, sigma, order, data.indptr, data.indices,
                                              users[sl], 0.05, 11, offset=1000 + lo)
        neg_o[sl] = nb
        oracle.step(Po, Qo, None, users[sl], pos[sl], nb, opt, k + 1, st, REG)
    got = neg.cpu().numpy()
    same = (got == neg_o).mean()
    # adaptive: an fp32 CDF threshold within rounding of a bin edge may pick the neighbouring factor
    assert same == 1.0 if sampler == 1 else same > 0.99, same
    if same == 1.0:
        assert agree(e.P.cpu().numpy(), Po, cfg), maxerr(e.P.cpu().numpy(), Po)
        assert agree(e.Q.cpu().numpy(), Qo, cfg), maxerr(e.Q.cpu().numpy(), Qo)


@pytest.mark.parametrize("opt_name", ["sgd", "nesterov", "adam_09", "adam_01", "rmsprop"])
@pytest.mark.parametrize("d", [8, 64, 128, 256, 1000])
def test_full_concurrency_one_virtual_batch_per_launch_is_exact(opt_name, d):
    """Chip-wide concurrency, heavy contention (4,096 triples on 60 users x 40 items per launch).
    With B >= n every triple of a launch belongs to ONE virtual step, and with the rows brought to
    "now" between launches every view is the pre-step row: whatever the interleaving of header
    CAS, joins and accumulator adds, each row must take exactly one optimizer step per launch with
    the summed gradient."""
    cfg = OPTS[opt_name]
    U, I, n, launches = 60, 40, 4096, 5
    P, Q, _, _, _, _, _ = rand_problem(U, I, d, 5, seed=d, B=8)
    b = np.linspace(-0.05, 0.05, I).astype(np.float32)
    rng = np.random.default_rng(d)
    e = make_engine(P, Q, b, REG)
    e.set_optimizer(**cfg)
    e.alloc_opt_state()
    Po, Qo, bo = P.copy(), Q.copy(), b.copy()
    st, opt = oracle_state(Po, Qo, bo), oracle_opt(cfg)
    for t in range(launches):
        users = rng.integers(0, U, n).astype(np.int32)
        pos = rng.integers(0, I, n).astype(np.int32)
        neg = distinct_neg(pos, rng.integers(1, I, n).astype(np.int32), I)
        e.train_stream_batched(dev(users), dev(pos), n, sampler=0, neg=dev(neg))
        e.flush_lazy()
        oracle.step(Po, Qo, bo, users, pos, neg, opt, t + 1, st, REG)
    assert e.step_count == launches
    tol = 1e-4  # fp32 atomics sum ~100 gradients per row in arbitrary order
    assert agree(e.P.cpu().numpy(), Po, cfg, tol), maxerr(e.P.cpu().numpy(), Po)
    assert agree(e.Q.cpu().numpy(), Qo, cfg, tol), maxerr(e.Q.cpu().numpy(), Qo)
    assert agree(e.item_bias.cpu().numpy(), bo, cfg, tol)


@pytest.mark.parametrize("opt_name", ["adam_09", "nesterov", "rmsprop"])
@pytest.mark.parametrize("d", [32, 128, 256])
def test_full_concurrency_closes_lose_and_double_nothing(opt_name, d):
    """The close-and-apply protocol under contention, isolated from view staleness: with lr = 0
    the tables never move, so every gradient is a function of the inputs alone and the optimizer
    STATE (momentum_buffer / exp_avg / exp_avg_sq / square_avg) after several back-to-back launches
    — steps closed by whichever triple first touches a row in the next launch, while its
    batch-mates are already adding to the other accumulator — must equal the oracle's: no gradient
    lost, none applied twice, none attributed to the wrong step."""
    cfg = dict(OPTS[opt_name], lr=0.0)
    U, I, n, launches = 60, 40, 4096, 6
    P, Q, _, _, _, _, _ = rand_problem(U, I, d, 5, seed=d + 1, B=8)
    rng = np.random.default_rng(d)
    e = make_engine(P, Q, None, REG)
    e.set_optimizer(**cfg)
    state = e.alloc_opt_state()
    Po, Qo = P.copy(), Q.copy()
    st, opt = oracle_state(Po, Qo, None), oracle_opt(cfg)
    for t in range(launches):
        users = rng.integers(1, U, n).astype(np.int32)
        pos = rng.integers(1, I, n).astype(np.int32)
        neg = distinct_neg(pos, rng.integers(1, I, n).astype(np.int32), I)
        e.train_stream_batched(dev(users), dev(pos), n, sampler=0, neg=dev(neg))  # no flush
        oracle.step(Po, Qo, None, users, pos, neg, opt, t + 1, st, REG)
    e.flush_lazy()
    assert np.array_equal(e.P.cpu().numpy(), P) and np.array_equal(e.Q.cpu().numpy(), Q)
    for k in ("mP", "vP", "mQ", "vQ"):
        if state[k] is not None:
            got, want = state[k].cpu().numpy(), st[k]
            scale = np.abs(want).max()
            assert scale > 0 and np.abs(got - want).max() <= 1e-4 * scale, (k, np.abs(got - want).max(), scale)


@pytest.mark.parametrize("opt_name", ["sgd", "nesterov", "adam_01", "rmsprop"])
def test_direct_steps_for_rows_alone_in_their_batch_are_the_deferred_steps(opt_name, monkeypatch):
    """r4: a user row that is alone in its virtual batch takes its step at once, under the lock its
    opener holds (k_valone + vs_contribute's direct path), instead of parking the gradient for the
    next visitor.  In the sequential limit that is the same arithmetic on the same operands: the
    tables after two launches equal BIT FOR BIT those of BPR_VS_DIRECT=0 (every row deferred) — with
    users that repeat inside a batch (not alone: deferred), users alone in one batch and repeated in
    the next, a pad user, and a launch boundary in between."""
    cfg = OPTS[opt_name]
    U, I, B, d = 5000, 300, 32, 128
    P, Q, _, _, _, _, _ = rand_problem(U, I, d, 10, seed=9, B=8)
    P *= 4
    Q *= 4
    b = np.linspace(-0.1, 0.1, I).astype(np.float32)
    rng = np.random.default_rng(3)
    n = 40 * B - 5
    users = rng.integers(1, U, n).astype(np.int32)  # 32 of 5,000: nearly all alone
    pos = rng.integers(1, I, n).astype(np.int32)
    neg = distinct_neg(pos, rng.integers(1, I, n).astype(np.int32), I)
    users[3] = users[9] = users[20]  # three of one user inside batch 0
    users[B + 1] = users[20]  # the same user alone in batch 1 (its batch-0 step is pending)
    users[2 * B + 4] = users[2 * B + 5]  # a pair in batch 2 ...
    users[5 * B + 7] = users[2 * B + 4]  # ... alone again in batch 5
    users[40] = 0
    out = {}
    for direct in ("1", "0"):
        monkeypatch.setenv("BPR_VS_DIRECT", direct)
        e = make_engine(P, Q, b, REG)
        e.set_optimizer(**cfg)
        e.alloc_opt_state()
        cut = 17 * B
        e.train_stream_batched(dev(users[:cut]), dev(pos[:cut]), B, sampler=0, neg=dev(neg[:cut]), max_inflight=1)
        e.train_stream_batched(dev(users[cut:]), dev(pos[cut:]), B, sampler=0, neg=dev(neg[cut:]), max_inflight=1)
        e.flush_lazy()
        out[direct] = (e.P.cpu().numpy().copy(), e.Q.cpu().numpy().copy(), e.item_bias.cpu().numpy().copy())
    for got, want in zip(out["1"], out["0"]):
        assert np.array_equal(got, want), maxerr(got, want)
    Po, Qo, bo = P.copy(), Q.copy(), b.copy()
    oracle_batches(Po, Qo, bo, users, pos, neg, B, cfg)
    assert np.abs(Po - P).max() > 1e-3
    assert agree(out["1"][0], Po, cfg), maxerr(out["1"][0], Po)
    assert agree(out["1"][1], Qo, cfg), maxerr(out["1"][1], Qo)


@pytest.mark.parametrize("direct", ["1", "0"])
def test_full_concurrency_many_batches_conserve_the_gradients(direct, monkeypatch):
    """The protocol at full concurrency over MANY virtual batches per launch (the shape of a real
    launch: most user rows alone in their batch, a few items under heavy contention).  With lr = 0
    the tables never move, so every gradient is a function of the inputs alone; with momentum
    0.99999 the momentum buffer is — to 1e-5 per step of misattribution, and 128 virtual steps are
    in flight here — the SUM of the row's gradients, whichever step a straggler's gradient was
    counted in — a late add that stays in an accumulator until the flush, 1,500 steps later, is off
    by 1.5 % of one gradient.  So buffer == oracle's to 2e-2 of the row's scale says: nothing lost,
    nothing applied twice, by either path (a user row holds about five gradients: one lost is 20 %)."""
    monkeypatch.setenv("BPR_VS_DIRECT", direct)
    cfg = dict(kind=1, lr=0.0, momentum=0.99999)
    U, I, B, d, n, launches = 20000, 40, 64, 128, 32768, 3
    P, Q, _, _, _, _, _ = rand_problem(U, I, d, 5, seed=4, B=8)
    rng = np.random.default_rng(8)
    e = make_engine(P, Q, None, REG)
    e.set_optimizer(**cfg)
    state = e.alloc_opt_state()
    Po, Qo = P.copy(), Q.copy()
    st = oracle_state(Po, Qo, None)
    for t in range(launches):
        users = rng.integers(1, U, n).astype(np.int32)
        pos = rng.integers(1, I, n).astype(np.int32)
        neg = distinct_neg(pos, rng.integers(1, I, n).astype(np.int32), I)
        e.train_stream_batched(dev(users), dev(pos), B, sampler=0, neg=dev(neg))
        oracle_batches(Po, Qo, None, users, pos, neg, B, cfg, t0=t * (n // B), st=st)
    e.flush_lazy()
    assert np.array_equal(e.P.cpu().numpy(), P) and np.array_equal(e.Q.cpu().numpy(), Q)
    for k in ("mP", "mQ"):
        got, want = state[k].cpu().numpy(), st[k]
        row_scale = np.abs(want).max(axis=1, keepdims=True)
        bad = np.abs(got - want) > 2e-2 * row_scale + 1e-7
        assert row_scale.max() > 0 and bad.sum() == 0, (k, int(bad.any(axis=1).sum()), "rows off")


@pytest.mark.parametrize("opt_name", ["adam_09", "nesterov", "sgd"])
def test_mode_switches_keep_one_trajectory(opt_name):
    """STRICT steps, then the batched stream, then STRICT again, with rows left pending / stale at
    every switch: one trajectory, equal to the oracle's."""
    cfg = OPTS[opt_name]
    U, I, d, B = 200, 150, 64, 16
    P, Q, _, _, _, _, _ = rand_problem(U, I, d, 5, seed=9, B=8)
    P *= 4
    Q *= 4
    rng = np.random.default_rng(3)
    n = 30 * B
    users = rng.integers(1, U, n).astype(np.int32)
    pos = rng.integers(1, I, n).astype(np.int32)
    neg = distinct_neg(pos, rng.integers(1, I, n).astype(np.int32), I)
    e = make_engine(P, Q, None, REG)
    e.set_optimizer(**cfg)
    e.alloc_opt_state()
    a, b2 = 10 * B, 20 * B
    e.train_strict(dev(users[:a]), dev(pos[:a]), B, sampler=0, neg=dev(neg[:a]))
    e.train_stream_batched(dev(users[a:b2]), dev(pos[a:b2]), B, sampler=0, neg=dev(neg[a:b2]),
                           max_inflight=1)
    e.train_strict(dev(users[b2:]), dev(pos[b2:]), B, sampler=0, neg=dev(neg[b2:]))
    e.flush_lazy()
    Po, Qo = P.copy(), Q.copy()
    oracle_batches(Po, Qo, None, users, pos, neg, B, cfg)
    assert e.step_count == 30
    assert agree(e.P.cpu().numpy(), Po, cfg), maxerr(e.P.cpu().numpy(), Po)
    assert agree(e.Q.cpu().numpy(), Qo, cfg), maxerr(e.Q.cpu().numpy(), Qo)


def test_hyper_parameter_change_applies_pending_steps_first():
    """set_optimizer with new values while steps are pending: the pending steps and the replay of
    the missed ones use the hyper-parameters in force when they were taken."""
    U, I, d, B = 100, 80, 32, 8
    P, Q, _, _, _, _, _ = rand_problem(U, I, d, 5, seed=4, B=8)
    rng = np.random.default_rng(6)
    n = 24 * B
    users = rng.integers(1, U, n).astype(np.int32)
    pos = rng.integers(1, I, n).astype(np.int32)
    neg = distinct_neg(pos, rng.integers(1, I, n).astype(np.int32), I)
    c1 = dict(kind=2, lr=0.01, betas=(0.9, 0.999))
    c2 = dict(kind=2, lr=0.002, betas=(0.5, 0.99))
    e = make_engine(P, Q, None, REG)
    e.set_optimizer(**c1)
    e.alloc_opt_state()
    h = 12 * B
    e.train_stream_batched(dev(users[:h]), dev(pos[:h]), B, sampler=0, neg=dev(neg[:h]), max_inflight=1)
    e.set_optimizer(**c2)
    e.train_stream_batched(dev(users[h:]), dev(pos[h:]), B, sampler=0, neg=dev(neg[h:]), max_inflight=1)
    e.flush_lazy()
    Po, Qo = P.copy(), Q.copy()
    st, _ = oracle_batches(Po, Qo, None, users[:h], pos[:h], neg[:h], B, c1)
    oracle_batches(Po, Qo, None, users[h:], pos[h:], neg[h:], B, c2, t0=12, st=st)
    assert agree(e.P.cpu().numpy(), Po, c1), maxerr(e.P.cpu().numpy(), Po)
    assert agree(e.Q.cpu().numpy(), Qo, c1), maxerr(e.Q.cpu().numpy(), Qo)


@pytest.mark.parametrize("sampler", [1, 2])
@pytest.mark.parametrize("seen", ["", "csr", "list"])
def test_full_concurrency_picks_match_the_oracle(sampler, seen, monkeypatch):
    """lr = 0 freezes the tables, so the negatives of a chip-wide launch must equal the oracle's
    draw for draw (Philox counters do not depend on the launch geometry)."""
    if seen:
        monkeypatch.setenv("BPR_SEEN", seen)
    from revisit_bpr.datasets import synthetic

    data = synthetic.generate(2000, 700, 60_000, median_per_user=20, seed=3)
    d = 128
    rng = np.random.default_rng(2)
    P = ((rng.random((data.num_users, d)) - 0.5) / d * 4).astype(np.float32)
    Q = ((rng.random((data.num_items, d)) - 0.5) / d * 4).astype(np.float32)
    P[0] = 0
    Q[0] = 0
    e = make_engine(P, Q, None, REG)
    e.bind_seen_csr(dev(data.indptr), dev(data.indices))
    e.set_optimizer(kind=2, lr=0.0, betas=(0.9, 0.999))
    e.alloc_opt_state()
    e.adaptive_refresh()
    pu, pi = e.shuffle_epoch(dev(data.users), dev(data.items), seed=5)
    neg = torch.zeros_like(pu)
    sc = torch.zeros(4, device="cuda")
    e.train_stream_batched(pu, pi, 256, sampler=sampler, neg=neg, adaptive_p=0.02, seed=7,
                           offset=123, scalars=sc)
    users = pu.cpu().numpy()
    if sampler == 1:
        want = oracle.sample_uniform(data.indptr, data.indices, data.num_items, users, 7, 123)
    else:
        QT, sigma = oracle.adaptive_stats(Q)
        want, _, _ = oracle.sample_adaptive(P, sigma, oracle.adaptive_order(QT), data.indptr,
                                            data.indices, users, 0.02, 7, offset=123)
    got = neg.cpu().numpy()
    assert int(sc[3]) == data.nnz
    assert (got == want).mean() >= (1.0 if sampler == 1 else 0.998)
    assert np.array_equal(e.P.cpu().numpy(), P) and np.array_equal(e.Q.cpu().numpy(), Q)


def test_shuffle_epoch_is_a_seeded_permutation():
    from revisit_bpr.engine import Engine

    e = Engine(torch.zeros(10, 4, device="cuda"), torch.zeros(10, 4, device="cuda"))
    for n in (1, 2, 3, 1000, 96_126, 1 << 16):
        u = torch.arange(n, dtype=torch.int32, device="cuda")
        i = (u * 7 + 3).to(torch.int32)
        a_u, a_i = e.shuffle_epoch(u, i, seed=1)
        b_u, _ = e.shuffle_epoch(u, i, seed=2)
        c_u, _ = e.shuffle_epoch(u, i, seed=1)
        assert torch.equal(torch.sort(a_u).values, u)  # a permutation ...
        assert torch.equal(a_i, (a_u * 7 + 3).to(torch.int32))  # ... applied to both columns
        assert torch.equal(a_u, c_u)  # seeded
        if n >= 1000:
            assert not torch.equal(a_u, b_u)
            # no long monotone stretch survives: neighbours are decorrelated
            assert float((a_u[1:] > a_u[:-1]).float().mean()) == pytest.approx(0.5, abs=0.05)


def test_full_concurrency_learns_at_full_size_contention():
    """A chip-wide launch on a small, popularity-skewed problem (hot rows close a step every few
    hundred nanoseconds): finite tables, exact triple count, loss well below ln 2 after an epoch,
    every row current after the flush (headers agree with the step counter through a second
    flush being a no-op)."""
    from revisit_bpr.datasets import synthetic

    data = synthetic.generate(5000, 1200, 400_000, median_per_user=40, seed=8)
    d = 128
    g = torch.Generator().manual_seed(1)
    P = ((torch.rand(data.num_users, d, generator=g) - 0.5) / d)
    Q = ((torch.rand(data.num_items, d, generator=g) - 0.5) / d)
    P[0] = 0
    Q[0] = 0
    from revisit_bpr import engine as eng

    e = eng.Engine(P.cuda(), Q.cuda())
    e.set_reg(*REG)
    e.bind_seen_csr(dev(data.indptr), dev(data.indices))
    e.set_optimizer(eng.OPT_ADAM, lr=0.002, betas=(0.1, 0.999))
    e.alloc_opt_state()
    u, i = dev(data.users), dev(data.items)
    sc = torch.zeros(4, device="cuda")
    for epoch in range(3):
        pu, pi = e.shuffle_epoch(u, i, seed=epoch)
        sc.zero_()
        e.adaptive_refresh()
        e.train_stream_batched(pu, pi, 256, sampler=2, adaptive_p=0.02, seed=3,
                               offset=epoch * data.nnz, scalars=sc)
    e.flush_lazy()
    assert int(sc[3]) == data.nnz
    assert float(sc[0] / sc[3]) < 0.9 * math.log(2.0)
    assert torch.isfinite(e.P).all() and torch.isfinite(e.Q).all()
    assert not e.P[0].any() and not e.Q[0].any()
    before = (e.P.clone(), e.Q.clone())
    e.flush_lazy()
    assert torch.equal(before[0], e.P) and torch.equal(before[1], e.Q)
    assert e.step_count == 3 * math.ceil(data.nnz / 256)
