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
    num = batch["item"].size(-1) if batch["item"].dim() > 1 else 1
    if num != 1:
        raise NotImplementedError("the device samplers draw one negative per positive (every "
                                  "reference config uses num = 1)")
    return num


class UniformSampler(Sampler):
    """Uniform over the items the user has not seen, never item 0 (reference: :15-37)."""

    def __init__(self, num_items: int, neg_gen: torch.Generator) -> None:
        self._neg_gen = neg_gen
        self._num_items = num_items
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
        return self._engine

    def sample(self, batch: dict[str, torch.Tensor]) -> torch.Tensor:
        _num(batch)
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
        neg = eng.sample_uniform(inv.to(torch.int32), seed=self._neg_gen.initial_seed(),
                                 offset=self._drawn)
        self._drawn += users.numel()
        return neg.to(torch.long).unsqueeze(-1)


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
        _num(batch)
        model = self._bpr()
        eng = model.engine()
        users = batch["user"].reshape(-1)
        if not getattr(model, "_has_csr", False):
            indptr, indices = csr_from_padded(users, batch["seen_items"], eng.U)
            eng.bind_seen_csr(indptr, indices)
        neg = eng.sample_adaptive(users, self._sampling_prob, seed=self._neg_gen.initial_seed(),
                                  offset=self._drawn)
        self._drawn += users.numel()
        if self._iteration_cnt % self._every == 0:
            self.update_stats()
        return neg.to(torch.long).unsqueeze(-1)

    @torch.no_grad()
    def update_stats(self) -> None:
        model = self._bpr()
        model.sync()  # lazy dense-optimizer replay: the snapshot must see current item rows
        model.engine().adaptive_refresh()
