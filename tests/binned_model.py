"""A numpy restatement of k_sort_binned's ALGORITHM (revisit-bpr_amd/csrc/bpr_refresh.hip) — test infrastructure:
the two-level interpolated rank (float32 arithmetic as the kernel has it), the equi-depth bins, the counting sort and
the ranking inside a bin.  It pins the argument the kernel's exactness rests on — the bin is a monotone function of
the key, equal keys share a bin — on the CPU; the HIP kernel itself is compared with the oracle's order in
tests/test_gpu_parity.py (`-m gpu`)."""
import numpy as np

F = np.float32


def bins_of(col: np.ndarray, n_bins: int):
    """bin per key (0 = the largest keys) or None when the column has no spread; plus the level-1 / level-2 pieces"""
    col = col.astype(F)
    n = len(col)
    vmin, vmax = F(col.min()), F(col.max())
    if not vmax > vmin:
        return None
    cmax, cscale = vmax, F(F(1024.0) / F(vmax - vmin))
    x = (cmax - col) * cscale
    cb = np.clip(x.astype(np.int32), 0, 1023)
    coarse = np.bincount(cb, minlength=1024)
    cum = np.concatenate([[0], np.cumsum(coarse)[:-1]])
    frac = np.clip(x - cb.astype(F), F(0), F(0.999))
    r = cum[cb].astype(F) + frac * coarse[cb].astype(F)
    crowded = np.nonzero(coarse > 32)[0]
    if len(crowded):
        h_lo, h_hi = int(crowded[0]), int(crowded[-1])
        inside = (cb >= h_lo) & (cb <= h_hi)
        ftop = F(cmax - F(h_lo) / cscale)
        fscale = F(cscale * F(F(1024.0) / F(h_hi - h_lo + 1)))
        x2 = (ftop - col) * fscale
        fb = np.clip(x2.astype(np.int32), 0, 1023)
        fine = np.bincount(fb[inside], minlength=1024)
        fcum = np.concatenate([[0], np.cumsum(fine)[:-1]])
        frac2 = np.clip(x2 - fb.astype(F), F(0), F(0.999))
        r2 = F(cum[h_lo]) + (fcum[fb].astype(F) + frac2 * fine[fb].astype(F))
        r = np.where(inside, r2, r)
    bscale = F(F(n_bins) / F(n))
    return np.clip((r.astype(F) * bscale).astype(np.int32), 0, n_bins - 1)


def orderable(col: np.ndarray) -> np.ndarray:
    b = col.astype(F).view(np.uint32).copy()
    b[b == 0x80000000] = 0  # -0 == +0
    neg = (b & 0x80000000) != 0
    return np.where(neg, ~b, b | np.uint32(0x80000000)).astype(np.uint32)


def binned_order(col: np.ndarray, n_bins: int = 8192, bin_max: int = 64):
    """the order k_sort_binned produces, or None when it would hand the column to the radix fallback"""
    bins = bins_of(col, n_bins)
    if bins is None:
        return None
    counts = np.bincount(bins, minlength=n_bins)
    if counts.max() > bin_max:
        return None
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    u = orderable(col)
    order = np.empty(len(col), np.int64)
    members = np.argsort(bins, kind="stable")  # any order inside a bin would do: the ranking below is what sorts
    at = 0
    for b in np.nonzero(counts)[0]:
        m = members[at:at + counts[b]]
        at += counts[b]
        for i in m:  # the members that precede i: larger key, or equal key and lower id
            rank = int(np.sum((u[m] > u[i]) | ((u[m] == u[i]) & (m < i))))
            order[starts[b] + rank] = i
    return order
