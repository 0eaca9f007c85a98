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
    """Keeps the replicated tensors (item table, optional item bias) consistent across ranks.

    On a ROCm device the two elementwise passes around the all-reduce are the fused kernels
    ``bpr_item_delta`` / ``bpr_item_fold`` of libbprcore (one pass over the table each, buffers
    allocated once); on CPU tensors (gloo tests) the same algebra runs as torch ops."""

    def __init__(self, tensors: list[torch.Tensor], group: Optional[dist.ProcessGroup] = None,
                 scale: float = 1.0) -> None:
        self.tensors = [t.detach() for t in tensors if t is not None]
        self.group = group
        self.scale = scale  # 1.0: apply every rank's update; 1/world: DDP-style mean
        self.base = [t.clone() for t in self.tensors]
        self._own = [torch.empty_like(t) for t in self.tensors]
        self._tot = [torch.empty_like(t) for t in self.tensors]
        self._pending = False
        self.timing = False  # record the all-reduce on the side stream with events (bench.py)
        self._events: list = []
        self._cuda = bool(self.tensors) and self.tensors[0].is_cuda
        self._side = torch.cuda.Stream() if self._cuda else None
        self._lib = None
        if self._cuda:
            from revisit_bpr import native

            self._lib, self._check = native.load(), native.check
            for t in self.tensors:
                if t.dtype != torch.float32 or not t.is_contiguous():
                    raise ValueError("ItemSync needs contiguous float32 tensors")

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def rank(self) -> int:
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def max_over_ranks(self, value: int) -> int:
        """MAX of a host integer over the group (e.g. chunks per epoch: ranks whose shard holds
        fewer chunks must still take part in every reconciliation, or the collectives of different
        ranks stop matching up)."""
        if self.world == 1:
            return int(value)
        dev = self.tensors[0].device if self.tensors else torch.device("cpu")
        t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    # ---- the two elementwise passes ---------------------------------------------------------
    def _delta(self, t, b, own, tot) -> None:
        if self._lib is not None:
            self._check(self._lib.bpr_item_delta(t.data_ptr(), b.data_ptr(), own.data_ptr(),
                                                 tot.data_ptr(), t.numel(),
                                                 torch.cuda.current_stream().cuda_stream))
        else:
            torch.sub(t, b, out=own)
            tot.copy_(own)

    def _fold(self, t, b, own, tot, rebase: bool) -> None:
        if self._lib is not None:
            self._check(self._lib.bpr_item_fold(t.data_ptr(), b.data_ptr(), own.data_ptr(),
                                                tot.data_ptr(), self.scale, int(rebase), t.numel(),
                                                torch.cuda.current_stream().cuda_stream))
        else:
            st = tot * self.scale
            b.add_(st)  # identical on every rank: the bases never drift apart
            if rebase:
                t.copy_(b)
            else:
                t.add_(st - own)

    def _all_reduce(self, tot) -> None:
        if self.world > 1:
            if self.timing and self._cuda:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)
                b.record()
                self._events.append((a, b, tot.numel() * tot.element_size()))
            else:
                dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)

    def timing_read(self) -> dict:
        """Average duration of the recorded all-reduces (ms, on the stream they ran on) and their
        message size; clears the record."""
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b, _ in self._events]
        size = self._events[0][2] if self._events else 0
        self._events = []
        return {"all_reduces": len(ms), "all_reduce_ms_avg": (sum(ms) / len(ms)) if ms else None,
                "message_bytes": size}

    # ---- API ----------------------------------------------------------------------------------
    @torch.no_grad()
    def sync(self) -> None:
        """Blocking reconcile: Q <- Q_base + scale * all_reduce(Q - Q_base), bit-identical on every
        rank."""
        self.start()
        self.finish(rebase=True)

    @torch.no_grad()
    def start(self) -> None:
        """Cut this rank's deltas on the compute stream (a consistent snapshot) and launch their
        all-reduce on the side stream."""
        if self._pending:
            self.finish()
        for t, b, own, tot in zip(self.tensors, self.base, self._own, self._tot):
            self._delta(t, b, own, tot)
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                for tot in self._tot:
                    self._all_reduce(tot)
        else:
            for tot in self._tot:
                self._all_reduce(tot)
        self._pending = True

    @torch.no_grad()
    def step(self) -> None:
        """finish() of the reconciliation in flight + start() of the next one, with the two
        elementwise passes fused into one (nothing trains in between)."""
        if not self._pending:
            return self.start()
        if self._lib is None:
            self.finish()
            return self.start()
        torch.cuda.current_stream().wait_stream(self._side)
        for t, b, own, tot in zip(self.tensors, self.base, self._own, self._tot):
            self._check(self._lib.bpr_item_fold_delta(t.data_ptr(), b.data_ptr(), own.data_ptr(),
                                                      tot.data_ptr(), self.scale, t.numel(),
                                                      torch.cuda.current_stream().cuda_stream))
        self._side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._side):
            for tot in self._tot:
                self._all_reduce(tot)

    @torch.no_grad()
    def finish(self, rebase: bool = False) -> None:
        """Fold the other ranks' contributions into the live replicas (one period late): the
        replica keeps what it learned since start(); the base becomes the reconciled cut.
        rebase=True (only valid when nothing trained since start()) sets replica = base."""
        if not self._pending:
            return
        self._pending = False
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        for t, b, own, tot in zip(self.tensors, self.base, self._own, self._tot):
            self._fold(t, b, own, tot, rebase)
