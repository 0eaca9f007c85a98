"""Streaming ranking-metric base classes (API of the reference's revisit_bpr/metrics/metric.py).

Every concrete metric keeps two running scalars — the sum of per-user scores and the number of
users — and reports their ratio; `compute` returns the per-user scores of one batch.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Optional

import torch


class Metric(ABC):
    """Base class of all metrics."""

    _accelerator = None

    @property
    def accelerator(self):
        return self._accelerator

    def set_accelerator(self, value) -> None:
        self._accelerator = value

    @abstractmethod
    def state_dict(self) -> dict[str, Any]: ...

    @abstractmethod
    def load_state_dict(self, state_dict: dict[str, Any]) -> None: ...

    @abstractmethod
    def __call__(self, output: torch.Tensor, target: torch.Tensor) -> None: ...

    @abstractmethod
    def compute(self, output: torch.Tensor, target: torch.Tensor) -> torch.Tensor: ...

    @abstractmethod
    def get_metric(self, reset: bool = False) -> torch.Tensor: ...

    @abstractmethod
    def reset(self) -> None: ...


class MaskedMetric(Metric):
    """Metrics whose inputs carry a validity mask."""

    @abstractmethod
    def __call__(self, output, target, mask: Optional[torch.Tensor] = None) -> None: ...

    @abstractmethod
    def compute(self, output, target, mask: Optional[torch.Tensor] = None) -> torch.Tensor: ...


class RunningMean:
    """The shared accumulator: total score / total users, stored under `<name>` / `total_count`."""

    def __init__(self, owner: Metric, key: str) -> None:
        self._owner, self.key = owner, key
        self.total = 0
        self.count = 0

    def _device(self):
        acc = self._owner.accelerator
        return acc.device if acc is not None else torch.device("cpu")

    def add(self, per_user: torch.Tensor, n_users: int, device) -> None:
        self.count = self.count + torch.tensor(n_users, device=device)
        self.total = self.total + per_user.sum()

    def value(self) -> torch.Tensor:
        return self.total / self.count

    def reset(self) -> None:
        dev = self._device()
        self.total = torch.tensor(0.0, device=dev)
        self.count = torch.tensor(0.0, device=dev)

    def state(self) -> dict[str, Any]:
        return {self.key: self.total, "total_count": self.count}

    def load(self, state: dict[str, Any]) -> None:
        self.total, self.count = state[self.key], state["total_count"]
        if self._owner.accelerator is not None:
            dev = self._owner.accelerator.device
            self.total, self.count = self.total.to(dev), self.count.to(dev)


def validate_metric_inputs(output: torch.Tensor, target: torch.Tensor) -> None:
    if output.size() != target.size():
        raise IndexError(f"Different sizes in output and target tensors: output - {output.size()}, "
                         f"target - {target.size()}.")
    if not (target.eq(0) | target.eq(1)).all():
        raise ValueError("Target contains values outside of 0 and 1.")


def prepare_target(output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """`target` re-ordered by descending `output` (full ranking; reference metric.py:110-113)."""
    return torch.gather(target, dim=-1, index=torch.argsort(-output, dim=-1))


def ranked_targets(output: torch.Tensor, target: torch.Tensor, k: int) -> torch.Tensor:
    """First k columns of prepare_target via top-k instead of a full sort of I scores."""
    k = min(k, output.size(-1))
    return torch.gather(target, dim=-1, index=torch.topk(output, k, dim=-1).indices)
