#!/usr/bin/env python
"""VERDICT r3 item 2 — "own the positive row": what every ordering of a STREAM chunk costs in
memory-side line-atomics, counted on the bench's own synthetic ML-20M-shaped chunk (CPU only).

k_stream today walks a chunk sorted by USER in runs of 8: the user row lives in registers (plain
store when the user is owned by one run, one atomic row-add per piece otherwise), both item rows
take one full-line fp32 atomic per 128 B.  A row has d*4/128 lines.  Alternatives counted here:
  item-major   chunk sorted by POSITIVE item, runs of 8: the positive row accumulates in registers
               (one atomic row-add per piece), the user row of every triple is an atomic add unless
               that user has a single triple in the whole chunk (then: plain read-modify-write)
  hybrid       per triple the cheaper of the two homes (lower bound: ignores that a run must be
               homogeneous)
  DSGD strata  W x W cells (user block x item block), W phases of W disjoint cells: a worker owns
               p_u AND q_i of its cell; population per cell and per phase at this chunk size.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "revisit-bpr_amd")]
from revisit_bpr.datasets import synthetic  # noqa: E402


def pieces(keys_sorted, L):
    """(#pieces, #triples in owned keys): a key whose triples all fall inside one run of L is owned
    (no atomics); every other key costs one atomic row-add per run it touches."""
    n = len(keys_sorted)
    run = np.arange(n) // L
    first = np.r_[True, keys_sorted[1:] != keys_sorted[:-1]]
    kid = np.cumsum(first) - 1
    # distinct (key, run) pairs
    pr = np.unique(kid.astype(np.int64) * (n // L + 2) + run)
    per_key = np.bincount((pr // (n // L + 2)).astype(np.int64), minlength=kid[-1] + 1)
    cut = per_key > 1
    return int(per_key[cut].sum()), cut, kid


def main():
    d, L = 128, 8
    lines = d * 4 // 128
    data = synthetic.generate_named("ml-20m", eval_users=10_000, seed=13)
    n, I, U = data.nnz, data.num_items, data.num_users
    chunk = int(I * np.log(I) / 256) * 256
    rng = np.random.default_rng(0)
    sel = rng.permutation(n)[:chunk]
    u, i = data.users[sel], data.items[sel]
    print(f"chunk of {chunk} triples out of {n} ({U - 1} users x {I - 1} items), d={d}: {lines} lines per row")
    cu = np.bincount(u, minlength=U)
    ci = np.bincount(i, minlength=I)
    tu, ti = cu[u], ci[i]
    print(f"triples per user in the chunk: mean {chunk / (cu > 0).sum():.2f} over {(cu > 0).sum()} users; "
          f"share of triples whose user is alone in the chunk {np.mean(tu == 1):.3f}, <=2: {np.mean(tu <= 2):.3f}")
    top = np.sort(ci)[::-1]
    print(f"triples per positive item: {(ci > 0).sum()} distinct rows; top 256 rows carry {top[:256].sum() / chunk:.3f}, "
          f"top 1024 {top[:1024].sum() / chunk:.3f}; share of triples on items with >= 8 triples {np.mean(ti >= 8):.3f}")
    # ---- (a) user-major, today
    o = np.argsort(u, kind="stable")
    pu, cut_u, _ = pieces(u[o], L)
    a_lines = lines * (2 + pu / chunk)
    print(f"\n(a) user-major runs of {L} (today):   neg {lines} + pos {lines} + user pieces {lines * pu / chunk:.2f}"
          f"  = {a_lines:.2f} line-atomics per triple")
    # ---- (b) item-major
    o = np.argsort(i, kind="stable")
    pi, cut_i, kid = pieces(i[o], L)
    owned_item_triples = 0  # items wholly inside a run still need ONE atomic row add: negatives of
    # other groups hit the same row concurrently, a plain store would lose them
    n_item_pieces = pi + int((~cut_i).sum())
    user_atomic = np.mean(tu > 1)
    b_lines = lines * (1 + n_item_pieces / chunk + user_atomic)
    print(f"(b) item-major runs of {L}:            neg {lines} + pos pieces {lines * n_item_pieces / chunk:.2f} "
          f"+ user {lines * user_atomic:.2f} (every triple whose user is not alone in the chunk)"
          f"  = {b_lines:.2f}")
    # ---- (c) hybrid lower bound: a triple goes item-major only when that saves lines
    pos_cost_item_major = lines / np.minimum(ti, L)          # its share of the piece's one row-add
    user_cost_item_major = np.where(tu > 1, lines, 0.0)
    cost_b = pos_cost_item_major + user_cost_item_major
    cost_a = lines + lines * (pu / chunk)                     # pos atomic + average user-piece share
    c_lines = lines + np.minimum(cost_a, cost_b).mean()
    print(f"(c) hybrid, per-triple best home (bound): {c_lines:.2f}   "
          f"(share of triples better off item-major: {np.mean(cost_b < cost_a):.3f})")
    rate = 9.5e9
    for name, v in (("a", a_lines), ("b", b_lines), ("c", c_lines)):
        print(f"    ({name}) at the unit rate of 9.5 G line-atomics/s: {v * chunk / rate * 1e3:.3f} ms per chunk")
    print("    measured today (profiles/r03_sweep_kstream_v1.txt): 0.2006 ms with given negatives — "
          "0.75 of that rate (popularity skew on the channels), 0.2205 ms with the adaptive sampler")
    # ---- (d) DSGD strata
    print("\n(d) DSGD strata on this chunk (W workers, W phases, cell = user block x item block, blocks "
          "balanced by triple count):")
    for W in (8, 32, 64, 256):
        ub = np.searchsorted(np.cumsum(np.bincount(u, minlength=U)) , np.arange(1, W) * chunk / W)
        ib_order = np.argsort(-ci)  # deal items round-robin by popularity: balanced item blocks
        iblock = np.empty(I, np.int64)
        iblock[ib_order] = np.arange(I) % W
        cell = np.searchsorted(ub, u, side="right") * W + iblock[i]
        pop = np.bincount(cell, minlength=W * W)
        phase_max = [pop.reshape(W, W)[np.arange(W), (np.arange(W) + k) % W].max() for k in range(W)]
        print(f"    W={W:4d}: {W * W:6d} cells, {pop.mean():8.1f} triples per cell (max {pop.max()}); "
              f"a launch = {W} phases with a grid barrier each, critical path = sum of the phases' largest "
              f"cells = {int(np.sum(phase_max))} triples in sequence per worker "
              f"(today a group walks {chunk // 7680 + 1} triples per launch)")
    print("    (the chip holds 7,680 groups: W = 256 workgroups of 32 groups is the natural mapping — 3 triples "
          "per cell and 256 grid barriers per launch; W = 8 (one worker per XCD) cannot own rows in registers "
          "or LDS, and device atomics do not execute in the XCD's L2: profiles/r03_pmc_memside.txt)")


if __name__ == "__main__":
    main()
