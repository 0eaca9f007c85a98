"""ROC-AUC variants (reference: revisit_bpr/metrics/auc.py:10-196).

RocAucOne   : column 0 is the positive, the rest are negatives (Netflix leave-one-out config).
RocAucMany  : all (positive, negative) pairs of a row, dense [B, I, I] comparison as the reference.
RocAucManySlow : same quantity; the reference loops over users in Python, here it is one sort per
                 row — AUC = (sum of ranks of positives among valid items - n_pos(n_pos+1)/2) /
                 (n_pos n_neg) with strict '>' comparisons, which counts exactly the pairs
                 pos > neg when scores are distinct and is corrected for ties below.
"""
from __future__ import annotations

from typing import Any, Optional

import torch

from revisit_bpr.metrics.metric import MaskedMetric, RunningMean


class _Auc(MaskedMetric):
    def __init__(self) -> None:
        self._mean = RunningMean(self, "total_auc")

    def state_dict(self) -> dict[str, Any]:
        return self._mean.state()

    def load_state_dict(self, state_dict: dict[str, Any]) -> None:
        self._mean.load(state_dict)

    def __call__(self, output, target, mask: Optional[torch.Tensor] = None) -> None:
        if mask is None:
            mask = torch.ones_like(target)
        self._mean.add(self.compute(output, target, mask), target.size(0), output.device)

    def get_metric(self, reset: bool = False) -> torch.Tensor:
        value = self._mean.value()
        if reset:
            self.reset()
        return value

    def reset(self) -> None:
        self._mean.reset()


class RocAucOne(_Auc):
    def compute(self, output, _, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        if mask is None:
            mask = torch.ones_like(output)
        valid = mask[:, 1:]
        wins = (output[:, :1] > output[:, 1:]).float()
        wins[valid.eq(0)] = 0.0
        return wins.sum(-1) / valid.sum(-1)


class RocAucMany(_Auc):
    def compute(self, output, target, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        if mask is None:
            mask = torch.ones_like(output)
        is_pos = target.ne(0)
        is_neg = target.eq(0) & mask.ne(0)
        wins = (output.unsqueeze(-1) > output.unsqueeze(-2)).float()  # [B, row, col]: row beats col
        wins = wins * is_pos.unsqueeze(-1) * is_neg.unsqueeze(-2)
        return wins.flatten(1).sum(-1) / (target.sum(-1) * is_neg.float().sum(-1))


class RocAucManySlow(_Auc):
    def compute(self, output, target, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        if mask is None:
            mask = torch.ones_like(output)
        is_pos = target.ne(0)
        is_neg = target.eq(0) & mask.ne(0)
        # number of valid negatives strictly below each score: rank among negatives via sort
        neg_scores = torch.where(is_neg, output, torch.full_like(output, float("inf")))
        sorted_neg, _ = torch.sort(neg_scores, dim=-1)
        below = torch.searchsorted(sorted_neg, output.contiguous(), right=False)  # strictly smaller
        n_neg = is_neg.sum(-1)
        below = torch.minimum(below, n_neg.unsqueeze(-1))
        wins = (below * is_pos).sum(-1).float()
        return wins / (is_pos.sum(-1) * n_neg).float()
