"""Top-k ranking metrics: NDCG, Recall, Precision, MAP, FBeta.

Arithmetic follows the reference exactly (revisit_bpr/metrics/ndcg.py:8-13,69-78;
recall.py:44-51; precision.py:44-51; map.py:45-66; fbeta.py:56-65) and is pinned to it by
tests/golden/metrics.npz; only the ranking step differs (one top-k instead of a full argsort).
"""
from __future__ import annotations

from typing import Any

import torch

from revisit_bpr.metrics.metric import Metric, RunningMean, ranked_targets, validate_metric_inputs


class _TopK(Metric):
    KEY = "total"

    def __init__(self, topk: int) -> None:
        assert topk > 0, f"Invalid topk value: {topk}"
        self._topk = topk
        self._mean = RunningMean(self, self.KEY)

    def state_dict(self) -> dict[str, Any]:
        return self._mean.state()

    def load_state_dict(self, state_dict: dict[str, Any]) -> None:
        self._mean.load(state_dict)

    def __call__(self, output: torch.Tensor, target: torch.Tensor) -> None:
        self._mean.add(self.compute(output, target), target.size(0), output.device)

    def get_metric(self, reset: bool = False) -> torch.Tensor:
        value = self._mean.value()
        if reset:
            self.reset()
        return value

    def reset(self) -> None:
        self._mean.reset()

    def _k(self, output: torch.Tensor) -> int:
        return min(output.size(-1), self._topk)


def _dcg(rel: torch.Tensor, gain: str) -> torch.Tensor:
    pos = torch.arange(rel.size(-1), dtype=torch.float, device=rel.device)
    if gain == "exp":
        return ((2 ** rel) - 1) / torch.log2(pos + 2.0)
    disc = 1 / (pos + 1.0)
    disc[0] = 1.0
    return rel * disc


class NDCG(_TopK):
    """Normalised discounted cumulative gain at k (gain 2^rel - 1 by default)."""

    KEY = "total_ndcg"

    def __init__(self, topk: int, gain_function: str = "exp") -> None:
        assert gain_function in ("exp", "linear"), f"Invalid gain_function value: {gain_function}"
        super().__init__(topk)
        self._gain = gain_function

    def compute(self, output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        k = self._k(output)
        got = _dcg(ranked_targets(output, target, k), self._gain).sum(-1)
        best = _dcg(torch.topk(target, k, dim=-1).values, self._gain).sum(-1)
        return torch.nan_to_num(got / best)


class Recall(_TopK):
    """hits@k / number of ALL relevant items of the user."""

    KEY = "total_recall"

    def compute(self, output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        validate_metric_inputs(output, target)
        hits = ranked_targets(output, target, self._k(output)).sum(-1)
        return torch.nan_to_num(hits / target.sum(-1))


class Precision(_TopK):
    KEY = "total_precision"

    def compute(self, output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        validate_metric_inputs(output, target)
        k = self._k(output)
        return ranked_targets(output, target, k).sum(-1) / k


class MAP(_TopK):
    """Mean average precision at k (normalised by min(#relevant, k) unless normalized=False)."""

    KEY = "total_map"

    def __init__(self, topk: int, normalized: bool = True) -> None:
        super().__init__(topk)
        self._normalized = normalized

    def compute(self, output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        validate_metric_inputs(output, target)
        k = self._k(output)
        rel = ranked_targets(output, target, k)
        ranks = torch.arange(1, k + 1, dtype=torch.long, device=output.device)
        prec_at_hits = rel.cumsum(-1) / ranks * rel
        denom = target.sum(-1).clamp(max=k) if self._normalized else rel.sum(-1)
        return torch.nan_to_num(prec_at_hits.sum(-1) / denom)


class FBeta(_TopK):
    KEY = "total_f"

    def __init__(self, topk: int, beta: float = 1.0) -> None:
        super().__init__(topk)
        self._beta = beta
        self._precision, self._recall = Precision(topk), Recall(topk)

    def state_dict(self) -> dict[str, Any]:
        return {**self._mean.state(), "precision": self._precision.state_dict(),
                "recall": self._recall.state_dict()}

    def load_state_dict(self, state_dict: dict[str, Any]) -> None:
        self._mean.load(state_dict)
        self._precision.load_state_dict(state_dict["precision"])
        self._recall.load_state_dict(state_dict["recall"])

    def compute(self, output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        p, r = self._precision.compute(output, target), self._recall.compute(output, target)
        b2 = self._beta ** 2
        return (1.0 + b2) * p * r / (b2 * p + r + 1e-13)
