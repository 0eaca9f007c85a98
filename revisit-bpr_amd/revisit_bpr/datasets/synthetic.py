"""Synthetic implicit-feedback data shaped like the reference's datasets (README.md:51-57 of the
reference: users x items / actions / median user and item counts).  No network, no real data: the
benchmark and the parity runs use these.

Generator (seeded, numpy only):
  * user activity n_u ~ log-normal with the dataset's median, clipped to [min_per_user, I/4],
    rescaled so that sum(n_u) ~= actions;
  * item popularity ~ Zipf-Mandelbrot w_r = (r + shift)^-skew over a random permutation of items;
  * each user draws n_u distinct items by popularity; ids start at 1 (0 is the pad row);
  * "user-split" hold-out (as the reference's ML-20M / MSD split): `eval_users` users keep 80 % of
    their items as fold-in (training + seen) and the other 20 % become the ranking targets.
"""
from __future__ import annotations

import hashlib
import os
from dataclasses import dataclass
from pathlib import Path

import numpy as np

# name -> (users, items, actions, median items per user, min per user)
SHAPES = {
    "cfg1-synth": (10_000, 5_000, 500_000, 40, 5),
    "netflix": (9_949, 4_825, 563_577, 27, 5),
    "ml-20m": (136_677, 20_108, 9_700_000, 37, 5),
    "msd": (571_355, 41_140, 32_500_000, 39, 20),
    "yelp": (252_616, 92_089, 2_200_000, 5, 3),
}


@dataclass
class Interactions:
    num_users: int  # table rows incl. pad row 0
    num_items: int
    users: np.ndarray  # int32 [nnz] training triples' users (sorted by user)
    items: np.ndarray  # int32 [nnz] positives
    indptr: np.ndarray  # int64 [num_users+1] seen-items CSR (== training interactions)
    indices: np.ndarray  # int32 [nnz] sorted per row
    eval_users: np.ndarray  # int32 [E]
    eval_indptr: np.ndarray  # int64 [E+1] held-out targets per eval user
    eval_items: np.ndarray  # int32

    @property
    def nnz(self) -> int:
        return int(self.users.shape[0])


def generate(users: int, items: int, actions: int, median_per_user: float = 37.0,
             min_per_user: int = 5, eval_users: int = 0, seed: int = 13, item_skew: float = 1.5,
             item_shift: float = 60.0, holdout: float = 0.2) -> Interactions:
    rng = np.random.default_rng(seed)
    U, I = users + 1, items + 1
    mean = actions / users
    sigma = np.sqrt(2.0 * np.log(max(mean / median_per_user, 1.05)))
    n_u = np.exp(np.log(median_per_user) + sigma * rng.standard_normal(users))
    n_u = np.clip(n_u, min_per_user, max(min_per_user, items // 4))
    n_u = np.clip(np.round(n_u * (actions / n_u.sum())), min_per_user, items // 2).astype(np.int64)
    # popularity
    w = (np.arange(1, items + 1, dtype=np.float64) + item_shift) ** (-item_skew)
    w /= w.sum()
    perm = rng.permutation(items) + 1  # rank r -> item id
    cdf = np.cumsum(w)
    # oversample with replacement, dedupe per user, top up users that fell short
    uid = np.repeat(np.arange(1, U, dtype=np.int64), n_u)
    draw = perm[np.minimum(np.searchsorted(cdf, rng.random(uid.shape[0])), items - 1)]
    key = np.unique(uid * I + draw)
    for _ in range(6):
        got = np.bincount(key // I, minlength=U)[1:]
        short = n_u - got
        if (short <= 0).all():
            break
        uid2 = np.repeat(np.arange(1, U, dtype=np.int64), np.maximum(short, 0) * 2)
        draw2 = perm[np.minimum(np.searchsorted(cdf, rng.random(uid2.shape[0])), items - 1)]
        key = np.unique(np.concatenate([key, uid2 * I + draw2]))
    # trim users that now exceed their target (keep a random subset)
    ku, ki = key // I, key % I
    order = np.lexsort((rng.random(key.shape[0]), ku))
    ku, ki = ku[order], ki[order]
    start = np.concatenate([[0], np.cumsum(np.bincount(ku, minlength=U))])[:-1]
    pos_in_user = np.arange(ku.shape[0]) - start[ku]
    keep = pos_in_user < np.concatenate([[0], n_u])[ku]
    ku, ki = ku[keep], ki[keep]
    # hold-out
    ev = np.sort(rng.choice(np.arange(1, U), size=min(eval_users, users), replace=False)) \
        if eval_users > 0 else np.zeros(0, np.int64)
    is_ev = np.zeros(U, bool)
    is_ev[ev] = True
    held = is_ev[ku] & (rng.random(ku.shape[0]) < holdout)
    # never hold out everything / nothing is fine for the metric code (nan_to_num path)
    tr_u, tr_i = ku[~held], ki[~held]
    o = np.lexsort((tr_i, tr_u))
    tr_u, tr_i = tr_u[o], tr_i[o]
    indptr = np.concatenate([[0], np.cumsum(np.bincount(tr_u, minlength=U))]).astype(np.int64)
    he_u, he_i = ku[held], ki[held]
    o = np.lexsort((he_i, he_u))
    he_u, he_i = he_u[o], he_i[o]
    cnt = np.bincount(he_u, minlength=U)[ev] if ev.size else np.zeros(0, np.int64)
    eval_indptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    return Interactions(
        num_users=U, num_items=I, users=tr_u.astype(np.int32), items=tr_i.astype(np.int32),
        indptr=indptr, indices=tr_i.astype(np.int32), eval_users=ev.astype(np.int32),
        eval_indptr=eval_indptr, eval_items=he_i.astype(np.int32))


_LATENT_VERSION = 1  # bump when generate_latent's draws change (names the cache files)
_ARRAYS = ("users", "items", "indptr", "indices", "eval_users", "eval_indptr", "eval_items")


def generate_latent(users: int, items: int, actions: int, factors: int = 8, strength: float = 1.5,
                    median_per_user: float = 20.0, min_per_user: int = 5, seed: int = 13,
                    item_skew: float = 1.0, item_shift: float = 30.0, holdout: float = 0.2,
                    eval_users: int = -1, cache_dir=None) -> Interactions:
    """Small sets with learnable structure (for nDCG parity runs): user u picks its n_u items
    without replacement with probability ∝ popularity_i · exp(strength · <z_u, y_i>) (Gumbel top-k),
    z, y ~ N(0, I_factors / factors).  O(users x items) time (blocked over users): the ML-20M-sized
    set takes a minute or more, so `cache_dir` keeps the arrays of one argument set in an .npz there
    (written under a temporary name and renamed: safe with several processes asking at once)."""
    if cache_dir is not None:
        key = repr((_LATENT_VERSION, users, items, actions, factors, strength, median_per_user, min_per_user, seed,
                    item_skew, item_shift, holdout, eval_users))
        path = Path(cache_dir) / f"bpr_latent_{hashlib.sha1(key.encode()).hexdigest()[:16]}.npz"
        if path.exists():
            try:
                with np.load(path) as z:
                    return Interactions(num_users=int(z["num_users"]), num_items=int(z["num_items"]),
                                        **{k: z[k] for k in _ARRAYS})
            except Exception:  # a damaged file: make it again
                pass
        data = generate_latent(users, items, actions, factors, strength, median_per_user, min_per_user, seed,
                               item_skew, item_shift, holdout, eval_users)
        tmp = path.with_name(f"{path.stem}.{os.getpid()}.tmp.npz")
        np.savez(tmp, num_users=data.num_users, num_items=data.num_items, **{k: getattr(data, k) for k in _ARRAYS})
        os.replace(tmp, path)
        return data
    rng = np.random.default_rng(seed)
    U, I = users + 1, items + 1
    mean = actions / users
    sigma = np.sqrt(2.0 * np.log(max(mean / median_per_user, 1.05)))
    n_u = np.exp(np.log(median_per_user) + sigma * rng.standard_normal(users))
    n_u = np.clip(np.round(n_u * (actions / n_u.sum())), min_per_user, items // 3).astype(np.int64)
    logpop = -item_skew * np.log(rng.permutation(items) + 1.0 + item_shift)
    Z = rng.standard_normal((users, factors)) / np.sqrt(factors)
    Y = rng.standard_normal((items, factors)) / np.sqrt(factors)
    # Gumbel top-k per user, in blocks of users so that memory stays O(block x items)
    ku = np.repeat(np.arange(1, U, dtype=np.int64), n_u)
    picks = []
    block = max(1, min(users, (64 << 20) // max(items, 1)))
    for lo in range(0, users, block):
        hi = min(lo + block, users)
        score = logpop[None, :] + strength * factors * (Z[lo:hi] @ Y.T) + rng.gumbel(size=(hi - lo, items))
        kmax = int(n_u[lo:hi].max())
        top = np.argpartition(-score, kmax - 1, axis=1)[:, :kmax]
        top = np.take_along_axis(top, np.argsort(-np.take_along_axis(score, top, axis=1), axis=1), axis=1)
        picks.extend(top[r, :n_u[lo + r]] + 1 for r in range(hi - lo))
    ki = np.concatenate(picks).astype(np.int64)
    n_eval = users if eval_users < 0 else min(eval_users, users)
    ev = np.sort(rng.choice(np.arange(1, U), size=n_eval, replace=False))
    is_ev = np.zeros(U, bool)
    is_ev[ev] = True
    held = is_ev[ku] & (rng.random(ku.shape[0]) < holdout)
    tr_u, tr_i = ku[~held], ki[~held]
    o = np.lexsort((tr_i, tr_u))
    tr_u, tr_i = tr_u[o], tr_i[o]
    indptr = np.concatenate([[0], np.cumsum(np.bincount(tr_u, minlength=U))]).astype(np.int64)
    he_u, he_i = ku[held], ki[held]
    o = np.lexsort((he_i, he_u))
    he_u, he_i = he_u[o], he_i[o]
    cnt = np.bincount(he_u, minlength=U)[ev]
    return Interactions(
        num_users=U, num_items=I, users=tr_u.astype(np.int32), items=tr_i.astype(np.int32),
        indptr=indptr, indices=tr_i.astype(np.int32), eval_users=ev.astype(np.int32),
        eval_indptr=np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64),
        eval_items=he_i.astype(np.int32))


def latent_factors(users: int, items: int, factors: int = 8, seed: int = 13) -> tuple[np.ndarray, np.ndarray]:
    """The latent factors behind `generate_latent(users, items, ..., factors=factors, seed=seed)`:
    Z [users + 1, factors], Y [items + 1, factors] (row 0 = the pad row, zero) — the same draws, replayed
    in the generator's order.  A model initialised from them starts above the untrained floor, which
    makes a short training prefix informative (tests/golden/make_golden_fullscale.py)."""
    rng = np.random.default_rng(seed)
    rng.standard_normal(users)   # n_u
    rng.permutation(items)       # popularity ranks
    Z = rng.standard_normal((users, factors)) / np.sqrt(factors)
    Y = rng.standard_normal((items, factors)) / np.sqrt(factors)
    return (np.vstack([np.zeros((1, factors)), Z]).astype(np.float32),
            np.vstack([np.zeros((1, factors)), Y]).astype(np.float32))


def generate_named(name: str, eval_users: int = 0, seed: int = 13, scale: float = 1.0,
                   **kw) -> Interactions:
    users, items, actions, med, mn = SHAPES[name]
    if scale != 1.0:
        users, actions = max(16, int(users * scale)), max(64, int(actions * scale))
    return generate(users, items, actions, med, mn, eval_users=eval_users, seed=seed, **kw)


def leave_one_out(data: Interactions, seed: int = 13) -> Interactions:
    """The Netflix protocol of the reference (configs/RQ1/ours.yaml.j2: `OnePosCollator` + `RocAucOne`,
    `skip_seen: false`): every user gives up ONE of its interactions as the evaluation positive, the rest
    train.  The user's seen set (the CSR) stays WHOLE — the reference's `seen_items` of a user holds the
    held-out item too (experiments/bpr/dataset.py:201-207: the positive is addressed as an index into it),
    so the samplers never draw it and the evaluation ranks it against the items outside the seen set."""
    rng = np.random.default_rng(seed)
    cnt = np.diff(data.indptr)[1:]
    pick = data.indptr[1:-1] + (rng.random(cnt.shape[0]) * cnt).astype(np.int64)
    pick = pick[cnt > 1]  # (a user with a single interaction keeps it)
    held = np.zeros(data.nnz, bool)
    held[pick] = True
    ev = data.users[pick]
    return Interactions(
        num_users=data.num_users, num_items=data.num_items, users=data.users[~held], items=data.items[~held],
        indptr=data.indptr, indices=data.indices, eval_users=ev.astype(np.int32),
        eval_indptr=np.arange(len(ev) + 1, dtype=np.int64), eval_items=data.items[pick].astype(np.int32))
