"""The algorithm of k_sort_binned (tests/binned_model.py: a numpy restatement with the kernel's float32 arithmetic) on
the CPU: the bin is a monotone function of the key and equal keys share a bin — so bins + ranking inside a bin ARE the
stable descending order — and the two-level histogram over [min, max] keeps every bin far below its 64-key limit on
the distributions the columns of a trained table take (the one-level, clipped histogram did not: profiles/
r05_binned_sort.md).  The HIP kernel is compared with the oracle's order in tests/test_gpu_parity.py."""
import numpy as np
import pytest

from binned_model import binned_order, bins_of


def columns(rng, n):
    yield "normal", rng.standard_normal(n) * 0.05
    yield "mixture", np.where(rng.random(n) < 0.7, rng.standard_normal(n) * 0.005, rng.standard_normal(n) * 0.02)
    yield "laplace", rng.laplace(0, 0.01, n)
    yield "student-t3", rng.standard_t(3, n) * 0.005
    yield "student-t1.5", rng.standard_t(1.5, n) * 0.005
    yield "outliers", np.where(rng.random(n) < 0.995, rng.standard_normal(n) * 0.01, rng.standard_normal(n) * 3.0)
    yield "exponential", rng.exponential(0.05, n)
    yield "shifted", rng.standard_normal(n) * 0.05 + 3.0
    yield "two-clusters", np.where(rng.random(n) < 0.5, rng.standard_normal(n) * 0.01 - 1.0, rng.standard_normal(n) * 0.01 + 1.0)
    c = rng.standard_normal(n) * 0.05
    c[5] = c[7]
    c[n // 2] = c[n // 2 + 3]
    c[11], c[13], c[17] = 0.0, -0.0, 0.0
    yield "few-ties", c


@pytest.mark.parametrize("n", [2048, 4801, 20108])
def test_bins_are_monotone_in_the_key_and_small(n):
    rng = np.random.default_rng(n)
    for name, col in columns(rng, n):
        col = col.astype(np.float32)
        bins = bins_of(col, 8192 if n > 10240 else 4096 if n > 6144 else 2048)
        by_key = np.argsort(-col, kind="stable")
        assert np.all(np.diff(bins[by_key]) >= 0), name  # larger key -> same or earlier bin
        same = col[by_key][1:] == col[by_key][:-1]
        assert np.all(bins[by_key][1:][same] == bins[by_key][:-1][same]), name  # equal keys share a bin
        assert np.bincount(bins).max() <= 24, (name, np.bincount(bins).max())  # (limit 64; two far clusters: 11)


@pytest.mark.parametrize("n", [2048, 9999])
def test_bins_plus_ranking_are_the_stable_descending_order(n):
    rng = np.random.default_rng(7 * n)
    for name, col in columns(rng, n):
        col = col.astype(np.float32)
        got = binned_order(col, 4096 if n > 6144 else 2048)
        if got is None:
            continue  # the radix fallback's case
        assert np.array_equal(got, np.argsort(-col, kind="stable")), name


def test_columns_without_spread_or_with_heavy_ties_fall_back():
    assert binned_order(np.full(3000, 0.25, np.float32), 2048) is None
    rng = np.random.default_rng(1)
    tied = (np.round(rng.standard_normal(20000) * 20) / 400).astype(np.float32)  # a few dozen distinct values
    assert binned_order(tied, 8192) is None
