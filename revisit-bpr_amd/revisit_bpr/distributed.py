"""Multi-GPU layout of the BPR hot path: users sharded, item table replicated.

A triple (u, i, j) touches one user row and two item rows.  Users are partitioned across ranks
(every triple of user u runs on owner(u)), so P shards never communicate.  Q (and item_bias) is
replicated and every rank trains on its own replica; replicas are reconciled periodically by an
all-reduce(SUM) of each rank's item-table DELTA since the last reconciliation (RCCL over xGMI via
``torch.distributed``, backend "nccl" on ROCm; "gloo" in the CPU tests):

    Q  <-  Q_base + sum_r (Q_r - Q_base)

which applies every rank's updates exactly once — the multi-GPU analogue of the single-GPU
asynchronous-SGD path with a staleness of one sync period.  The reference has no working
multi-device path to mirror (DDP launcher exists, no config enables it).

``ItemSync.sync()`` is blocking; ``start()`` / ``finish()`` split it so the all-reduce runs on a
side stream under the next chunk's kernels.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.distributed as dist


def balanced_user_shards(indptr: np.ndarray, world: int) -> np.ndarray:
    """Contiguous user ranges balanced by interaction count (not user count).
    Returns bounds [world+1] with bounds[0] = 0 and bounds[-1] = U: rank r owns users
    [bounds[r], bounds[r+1])."""
    U = indptr.shape[0] - 1
    total = int(indptr[-1])
    bounds = [0]
    for r in range(1, world):
        target = total * r // world
        b = int(np.searchsorted(indptr, target, side="left"))
        bounds.append(min(max(b, bounds[-1]), U))
    bounds.append(U)
    return np.asarray(bounds, np.int64)


def owner_of(users: np.ndarray, bounds: np.ndarray) -> np.ndarray:
    return (np.searchsorted(bounds, users, side="right") - 1).astype(np.int32)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class ItemSync:
    """Keeps the replicated tensors (item table, optional item bias) consistent across ranks."""

    def __init__(self, tensors: list[torch.Tensor], group: Optional[dist.ProcessGroup] = None,
                 scale: float = 1.0) -> None:
        self.tensors = [t for t in tensors if t is not None]
        self.group = group
        self.scale = scale  # 1.0: apply every rank's update; 1/world: DDP-style mean
        self.base = [t.detach().clone() for t in self.tensors]
        self._pending = None
        self._side = torch.cuda.Stream() if self.tensors and self.tensors[0].is_cuda else None

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def sync(self) -> None:
        """Blocking reconcile: Q <- Q_base + scale * all_reduce(Q - Q_base)."""
        if self._pending is not None:
            self.finish()
        for t, b in zip(self.tensors, self.base):
            delta = t.detach() - b
            if self.world > 1:
                dist.all_reduce(delta, op=dist.ReduceOp.SUM, group=self.group)
            if self.scale != 1.0:
                delta.mul_(self.scale)
            with torch.no_grad():
                t.copy_(b + delta)
                b.copy_(t)

    def start(self) -> None:
        """Snapshot the replicas and launch the all-reduce of their deltas on the side stream."""
        if self._pending is not None:
            self.finish()
        snaps, deltas = [], []
        for t, b in zip(self.tensors, self.base):
            s = t.detach().clone()  # on the compute stream: a consistent cut of this rank's replica
            snaps.append(s)
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream())
        ctx = torch.cuda.stream(self._side) if self._side is not None else _null()
        with ctx:
            for s, b in zip(snaps, self.base):
                own = s - b
                tot = own.clone()
                if self.world > 1:
                    dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)
                deltas.append((own, tot))
        self._pending = (snaps, deltas)

    def finish(self) -> None:
        """Fold the other ranks' contributions into the live replicas (one period late)."""
        if self._pending is None:
            return
        snaps, deltas = self._pending
        self._pending = None
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        with torch.no_grad():
            for t, b, s, (own, tot) in zip(self.tensors, self.base, snaps, deltas):
                if self.scale != 1.0:
                    # scaled total replaces this rank's own (unscaled) contribution as well
                    others = tot * self.scale - own
                else:
                    others = tot - own
                t.add_(others)          # live replica keeps what it learned since the snapshot
                b.copy_(s + others)     # new base = the reconciled cut
