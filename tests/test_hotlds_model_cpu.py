"""The algorithm of the LDS tier of k_stream's hot block (tests/hotlds_model.py: a numpy restatement of the run
partition, the deal of runs to persistent workgroups and the private-delta algebra) on the CPU:

  * the zones partition [0, n) exactly — every triple in exactly one run, runs in order, zone boundaries on whole
    wave-loads, run lengths L / L/2 / L/4 — for every run length the API accepts and any tail share;
  * the deal hands every run to exactly one workgroup, wave-load by wave-load, a workgroup's own runs in rising order
    (so its short runs come last), consecutive wave-loads on different workgroups;
  * one workgroup == sequential SGD; with many workgroups nothing is lost whatever the interleaving (the table after
    the launch is the start plus the sum of every triple's update), a workgroup always sees its OWN updates of a hot
    row, and what it misses of the others' is bounded by their updates of that launch.
The HIP kernel is held to the oracle in tests/test_gpu_hotlds.py."""
import numpy as np
import pytest

from hotlds_model import deal, run_launch, runs_of, zones


@pytest.mark.parametrize("L", [1, 2, 3, 4, 5, 8, 12, 30])
@pytest.mark.parametrize("gpw", [1, 2])
def test_zones_partition_the_launch(L, gpw):
    rng = np.random.default_rng(L * 10 + gpw)
    for n in [1, 7, 16, 199_168, 40_928] + [int(x) for x in rng.integers(1, 60_000, 6)]:
        for tail in (0, 12, 25, 50):
            t1, t2 = zones(n, L, gpw, tail)
            assert 0 <= t1 <= t2 <= n and t1 % (L * gpw) == 0 and (t2 - t1) % (max(1, L // 2) * gpw) == 0
            runs, R1, R2 = runs_of(n, L, t1, t2)
            assert R1 % gpw == 0 and (R2 - R1) % gpw == 0  # a wave's runs share a zone: wave-uniform run length
            flat = np.concatenate([np.arange(a, b) for a, b in runs]) if runs else np.zeros(0, int)
            assert np.array_equal(flat, np.arange(n)), (n, L, gpw, tail)  # every triple once, in order
            lens = np.array([b - a for a, b in runs])
            assert np.all(lens[:R1] == L) and np.all(lens[R1:R2] == max(1, L // 2))
            assert np.all(lens[R2:-1] == max(1, L // 4)) and 1 <= lens[-1] <= max(1, L // 4 if R2 < len(runs) else L)
            if tail == 0:
                assert t1 == t2 and n - t1 < L * gpw  # only the ragged end goes in short runs


def test_deal_is_a_partition_with_the_short_runs_last():
    n, L, gpw, grid = 199_168, 8, 2, 224
    t1, t2 = zones(n, L, gpw, 12)
    runs, R1, R2 = runs_of(n, L, t1, t2)
    per = deal(len(runs), grid, gpw)
    allr = np.sort(np.concatenate([np.asarray(p) for p in per]))
    assert np.array_equal(allr, np.arange(len(runs)))
    sizes = [sum(runs[r][1] - runs[r][0] for r in p) for p in per]
    assert max(sizes) - min(sizes) <= 2 * L * gpw  # a workgroup's share: one wave-load of full runs at most apart
    for b, p in enumerate(per):
        assert p == sorted(p) and p[0] == b * gpw  # rising: the zones of short runs come last
        lens = [runs[r][1] - runs[r][0] for r in p]
        assert all(x >= y for x, y in zip(lens[:-1], lens[1:]))  # never a longer run after a shorter one
        for a, c in zip(p[:-gpw:gpw], p[gpw::gpw]):
            assert c - a == grid * gpw  # consecutive wave-loads of the launch sit on different workgroups


def test_private_deltas_lose_nothing_and_one_workgroup_is_sequential():
    rng = np.random.default_rng(3)
    I, n, L, gpw = 60, 5_000, 8, 2
    rows = (rng.zipf(1.3, n) % I).astype(int)
    hot = np.zeros(I, bool)
    hot[np.argsort(-np.bincount(rows, minlength=I))[:12]] = True
    q0 = rng.normal(0, 1, I)
    lr = 0.01
    grads = lambda v, t: -lr * (0.3 * v + np.sin(t))  # depends on the value read: staleness would show
    # one workgroup: exactly sequential SGD over the triples in order
    seq = q0.astype(np.float64).copy()
    for t in range(n):
        seq[rows[t]] += grads(seq[rows[t]], t)
    one, _ = run_launch(q0, rows, grads, 1, gpw, L, 12, hot)
    assert np.allclose(one, seq, rtol=0, atol=1e-12)
    # many workgroups, any interleaving: the table is the start plus the sum of every triple's update
    const = lambda v, t: 1.0 + 0.0 * v
    for order in ("round-robin", "random", "one-by-one"):
        many, _ = run_launch(q0, rows, const, 16, gpw, L, 12, hot, order=order, seed=5)
        assert np.allclose(many - q0, np.bincount(rows, minlength=I).astype(float), rtol=0, atol=1e-9), order
    # a workgroup sees its own updates of a hot row: with ONE hot row touched by one workgroup only the reads are sequential
    rows2 = np.concatenate([np.full(16, 7), rng.integers(20, 40, n - 16)])  # the first wave-load (workgroup 0) hits row 7
    hot2 = np.zeros(I, bool)
    hot2[7] = True
    _, reads = run_launch(np.zeros(I), rows2, const, 16, gpw, L, 0, hot2, order="random", seed=1)
    assert np.array_equal(reads[:16], np.arange(16.0))
    # what a workgroup misses of the others' updates of a hot row is at most their updates of this launch
    rows3 = np.full(n, 5)
    hot3 = np.zeros(I, bool)
    hot3[5] = True
    _, reads = run_launch(np.zeros(I), rows3, const, 16, gpw, L, 12, hot3, order="random", seed=2)
    assert reads.min() == 0.0 and reads.max() <= n / 16 + 2 * L * gpw  # each workgroup counts only its own ~n / 16
