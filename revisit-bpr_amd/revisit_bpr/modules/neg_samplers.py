"""Negative samplers behind the reference's API (revisit_bpr/modules/neg_samplers.py:9-141), drawn
on the GPU by libbprcore instead of a [B, I] weight matrix + multinomial / argsort per batch.

Randomness: the reference consumes a ``torch.Generator``; here the generator only supplies the seed
(``neg_gen.initial_seed()``) of a counter-based Philox stream whose counter is the number of
negatives drawn so far, so draws are reproducible and independent of batch boundaries.  The streams
are different from torch's — parity with the reference is distributional.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Optional

import torch


class Sampler(ABC):
    @abstractmethod
    def sample(self, batch: dict[str, torch.Tensor]) -> torch.Tensor:
        """LongTensor [B, num] of negatives, num = batch["item"].size(-1)."""


def unique_seen_rows(users: torch.Tensor, seen_items: torch.Tensor):
    """Per distinct user of the batch (ascending id): its seen items, sorted, 0-padding and
    duplicates removed.  Returns (uniq [n], inv [B], counts [n], indices int32 [sum counts])."""
    uniq, inv = torch.unique(users, return_inverse=True)
    B = users.numel()
    first = torch.full((uniq.numel(),), B, dtype=torch.long, device=users.device)
    first.scatter_reduce_(0, inv, torch.arange(B, device=users.device), reduce="amin")
    rows, _ = torch.sort(seen_items[first], dim=1)
    keep = rows.ne(0)
    keep[:, 1:] &= rows[:, 1:].ne(rows[:, :-1])  # drop duplicates inside a row
    return uniq, inv, keep.sum(1), rows[keep].to(torch.int32)


def csr_from_padded(users: torch.Tensor, seen_items: torch.Tensor, num_users: int):
    """Seen-items CSR over all `num_users` users (int64 indptr [num_users+1], int32 indices) built on
    device from a batch's 0-padded ``seen_items`` [B, S]; users absent from the batch get empty rows."""
    uniq, _, counts, indices = unique_seen_rows(users, seen_items)
    per_user = torch.zeros(num_users + 1, dtype=torch.int64, device=users.device)
    per_user[uniq + 1] = counts
    return torch.cumsum(per_user, 0), indices


def _num(batch) -> int:
    """Negatives per row: num = batch["item"].size(-1) (reference: neg_samplers.py:32,76)."""
    return batch["item"].size(-1) if batch["item"].dim() > 1 else 1


def _distinct_rows(neg: torch.Tensor, redraw) -> torch.Tensor:
    """multinomial(..., num_samples=num) draws WITHOUT replacement (neg_samplers.py:33-37): the
    device sampler's independent draws are repeated where a row holds the same item twice
    (rejection — the accepted rows are uniform over the sets of `num` distinct unseen items)."""
    for _ in range(64):
        srt, _ = torch.sort(neg, dim=1)
        bad = (srt[:, 1:] == srt[:, :-1]).any(dim=1)
        if not bool(bad.any()):
            return neg
        neg[bad] = redraw(bad)
    raise RuntimeError("could not draw distinct negatives (fewer unseen items than negatives per row?)")


class UniformSampler(Sampler):
    """Uniform over the items the user has not seen, never item 0 (reference: :15-37).

    item_weights (extension; the reference's experiment class carries the same thing as
    `_item_counts ** neg_sampling_alpha`, experiments/bpr/exp.py:85-91, 282-293): [num_items]
    non-negative weights — a negative is drawn with probability w_i / (sum of w over the unseen)."""

    def __init__(self, num_items: int, neg_gen: torch.Generator,
                 item_weights: Optional[torch.Tensor] = None) -> None:
        self._neg_gen = neg_gen
        self._num_items = num_items
        self._item_weights = item_weights
        self._drawn = 0
        self._engine = None
        self._cap = 0

    def _ensure(self, n_users: int, device):
        from revisit_bpr.engine import Engine

        if self._engine is None or n_users > self._cap or self._engine.device != device:
            self._cap = max(1024, n_users)
            # only the row counts matter: the sampler kernels never read the tables
            self._engine = Engine(torch.zeros(self._cap, 1, device=device),
                                  torch.zeros(self._num_items, 1, device=device))
            if self._item_weights is not None:
                self._engine.bind_item_weights(self._item_weights)
        return self._engine

    def sample(self, batch: dict[str, torch.Tensor]) -> torch.Tensor:
        num = _num(batch)
        users, seen = batch["user"].reshape(-1), batch["seen_items"]
        if not users.is_cuda:
            raise RuntimeError("UniformSampler draws on the GPU: move the batch to the ROCm device")
        eng = self._ensure(users.numel(), users.device)
        # batch-local user ids 0..n_unique-1 index a batch-local CSR
        uniq, inv, counts, indices = unique_seen_rows(users, seen)
        local_ptr = torch.zeros(self._cap + 1, dtype=torch.int64, device=users.device)
        local_ptr[1:uniq.numel() + 1] = torch.cumsum(counts, 0)
        local_ptr[uniq.numel() + 1:] = local_ptr[uniq.numel()]
        eng.bind_seen_csr(local_ptr, indices)
        rows = inv.to(torch.int32)

        def draw(sel: torch.Tensor) -> torch.Tensor:  # `num` draws for the selected batch rows
            r = rows[sel].repeat_interleave(num)
            out = eng.sample_uniform(r, seed=self._neg_gen.initial_seed(), offset=self._drawn)
            self._drawn += r.numel()
            return out.to(torch.long).view(-1, num)

        neg = draw(torch.ones_like(rows, dtype=torch.bool))
        return neg if num == 1 else _distinct_rows(neg, draw)


class AdaptiveSampler(Sampler):
    """Adaptive oversampling of Rendle & Freudenthaler (2014) (reference: :40-132): factor
    f ~ |p_uf| sigma_f, rank r ~ Geometric(sampling_prob) clamped to #unseen, orientation by
    sign(p_uf), negative = r-th unseen item of the snapshot order of factor f; the snapshot is
    refreshed every `every` calls."""

    def __init__(self, model, num_items: int, sampling_prob: float, neg_gen: torch.Generator,
                 every: int) -> None:
        self._model = model
        self._num_items = num_items
        self._sampling_prob = sampling_prob
        self._neg_gen = neg_gen
        self._every = every
        self._iteration_cnt = 0
        self._drawn = 0

    def _bpr(self):
        model = self._model
        while hasattr(model, "module") and not hasattr(model, "logits_model"):
            model = model.module  # DDP / accelerate wrappers
        return model

    def sample(self, batch: dict[str, torch.Tensor]) -> torch.Tensor:
        self._iteration_cnt += 1
        num = _num(batch)
        model = self._bpr()
        eng = model.engine()
        users = batch["user"].reshape(-1)
        if not getattr(model, "_has_csr", False):
            indptr, indices = csr_from_padded(users, batch["seen_items"], eng.U)
            eng.bind_seen_csr(indptr, indices)
        if num == 1:
            neg = eng.sample_adaptive(users, self._sampling_prob, seed=self._neg_gen.initial_seed(),
                                      offset=self._drawn).to(torch.long).unsqueeze(-1)
            self._drawn += users.numel()
        else:
            neg = self._sample_many(model, eng, users, num)
        if self._iteration_cnt % self._every == 0:
            self.update_stats()
        return neg

    @torch.no_grad()
    def _sample_many(self, model, eng, users: torch.Tensor, num: int) -> torch.Tensor:
        """num > 1 negatives per row, literally as the reference (neg_samplers.py:84-121):
        `num` DISTINCT factors per row (multinomial without replacement over |p_uf| sigma_f), an
        independent Geometric rank for each, orientation by the sign of p_uf, and the rank-th
        unseen item of that factor's snapshot order (`bpr_adaptive_pick`).  Factors and ranks come
        from torch's generator here; no reference config uses num > 1."""
        model.sync()  # the live user rows, as of now
        p_u = model.logits_model.get_features()["user"].detach()[users]
        sigma = eng.adaptive_sigma()
        indptr = eng._keep["csr"][0]
        n_unseen = (eng.I - 1 - (indptr[users + 1] - indptr[users])).unsqueeze(-1)
        factor = torch.multinomial(p_u.abs() * sigma, num_samples=num, generator=self._neg_gen)
        rank = torch.empty_like(factor).geometric_(self._sampling_prob, generator=self._neg_gen)
        rank = torch.minimum(rank, n_unseen)
        rank = torch.where(p_u.gather(-1, factor).gt(0), rank - 1, n_unseen - rank)
        rank = rank.clamp(min=torch.zeros_like(n_unseen), max=n_unseen - 1)
        neg = eng.adaptive_pick(users.repeat_interleave(num), factor.reshape(-1), rank.reshape(-1))
        return neg.to(torch.long).view(-1, num)

    @torch.no_grad()
    def update_stats(self) -> None:
        model = self._bpr()
        model.sync()  # lazy dense-optimizer replay: the snapshot must see current item rows
        model.engine().adaptive_refresh()
