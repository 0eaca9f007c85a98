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

Two tiers (r4).  With a full refresh period per RANK and launch, N replicas each pour a whole
launch into the same few popular rows before anybody sees the others' updates — at aggressive
learning rates that diverges (profiles/r03_cadence_study.txt).  ``ItemSync(hot_rows=H, engine=...)``
therefore reconciles the H most popular rows of the WHOLE training set on their own: their updates
already live in the engine's hot delta block (a launch's ``hot_delta``), and ``hot_step()`` — after
every launch or sub-launch — all-reduces that block (H·d floats: 128 KB at H = 256, d = 128) and
folds the sum one launch later, while the cold rows keep the per-period delta all-reduce above
(``step()``).  ``bpr_hot_exchange`` (include/bprcore.h) is the fused pass.

The collective is pluggable: ``torch.distributed`` (RCCL / gloo), or ``LocalWorld`` — N ranks in ONE
process on one GPU, stepped round-robin, whose "all-reduce" is resolved when the last rank has
contributed: the same data flow as N processes (the protocol only ever reads a sum one step after
it was launched), without N processes.  tools/cadence_study.py runs its sweeps on it.
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


class _DistComm:
    """torch.distributed: the all-reduce is enqueued on a side stream, `complete` makes the current
    stream wait for it."""

    def __init__(self, group=None, cuda: bool = False, force: bool = False, split_bytes: int = 0) -> None:
        """force: enqueue the collectives even when the group has ONE rank (bench.py --force-dist: the
        N > 1 code path through RCCL on one GPU).  split_bytes > 0: an all-reduce of at least that many
        bytes runs as reduce-scatter + all-gather — on the 7-link xGMI ring each phase moves 1/N of
        the message per link and the two can be scheduled around the hot tier's small exchanges
        (NCCL / RCCL groups only, gloo keeps the plain all-reduce)."""
        self.group = group
        self.force = bool(force)
        self.split_bytes = int(split_bytes)
        self._side = torch.cuda.Stream() if cuda else None
        self._done: dict = {}  # tag -> event behind the last all-reduce launched under that tag
        self.timing = False
        self.events: list = []

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def rank(self) -> int:
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def launch(self, tensors, tag=None) -> None:
        if self.world <= 1 and not self.force:
            return
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                for t in tensors:
                    self._one(t, tag)
                ev = self._done.get(tag)
                if ev is None:
                    ev = self._done[tag] = torch.cuda.Event()
                ev.record(self._side)
        else:
            for t in tensors:
                self._one(t, tag)

    def _reduce(self, t) -> None:
        nbytes = t.numel() * t.element_size()
        w = self.world
        if self.split_bytes > 0 and nbytes >= self.split_bytes and t.is_cuda and dist.get_backend(self.group) == "nccl":
            # reduce-scatter + all-gather over a zero-padded flat view (numel need not divide by world)
            flat = t.view(-1)
            per = -(-flat.numel() // w)
            if per * w != flat.numel():
                buf = torch.zeros(per * w, dtype=t.dtype, device=t.device)
                buf[:flat.numel()].copy_(flat)
            else:
                buf = flat
            part = torch.empty(per, dtype=t.dtype, device=t.device)
            dist.reduce_scatter_tensor(part, buf, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_gather_into_tensor(buf, part, group=self.group)
            if buf is not flat:
                flat.copy_(buf[:flat.numel()])
            self.split_used = getattr(self, "split_used", 0) + 1
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def _one(self, t, tag) -> None:
        if self.timing and self._side is not None:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            self._reduce(t)
            b.record()
            self.events.append((a, b, t.numel() * t.element_size(), tag))
        else:
            self._reduce(t)

    def complete(self, tensors, tag=None) -> None:
        """The current stream waits for the exchange launched last under `tag` — not for the whole side
        stream: a hot exchange is folded without waiting for a cold all-reduce launched after it."""
        if self._side is not None:
            ev = self._done.get(tag)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
            elif self.world > 1 or self.force:
                torch.cuda.current_stream().wait_stream(self._side)

    def all_reduce_now(self, t, op=dist.ReduceOp.SUM) -> None:
        if self.world > 1 or self.force:
            dist.all_reduce(t, op=op, group=self.group)


class LocalWorld:
    """N ranks in one process.  `member(r)` is rank r's collective backend: `launch` registers the
    rank's buffers under a running sequence number, `complete` — called one protocol step later, when
    every rank has launched — replaces them by the element-wise sum over the ranks.  Ranks must be
    stepped round-robin (all ranks finish step k before any starts step k + 2)."""

    def __init__(self, world: int) -> None:
        self.world = int(world)
        self._pending: dict = {}   # (tag, seq) -> {rank: [tensors]}
        self._sums: dict = {}      # (tag, seq) -> [summed tensors], readers left

    def member(self, rank: int) -> "_LocalComm":
        return _LocalComm(self, rank)

    # The ranks may work on different HIP streams (every trainer of the overlapped schedule has its
    # own CU-masked pair): contributions and sums carry events, readers wait for them.
    @staticmethod
    def _mark(tensors):
        if tensors and tensors[0].is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(tensors[0].device))
            return ev
        return None

    @staticmethod
    def _wait(ev, tensors) -> None:
        if ev is not None:
            torch.cuda.current_stream(tensors[0].device).wait_event(ev)

    def _launch(self, rank, key, tensors) -> None:
        self._pending.setdefault(key, {})[rank] = (list(tensors), self._mark(tensors))

    def _complete(self, rank, key, tensors) -> None:
        if key not in self._sums:
            parts = self._pending.pop(key, {})
            if len(parts) != self.world:
                raise RuntimeError(f"LocalWorld: exchange {key} read by rank {rank} before ranks "
                                   f"{sorted(set(range(self.world)) - set(parts))} contributed "
                                   "(ranks must be stepped round-robin)")
            for r in range(self.world):
                self._wait(parts[r][1], tensors)
            # the sum in rank order, the same on every reader (RCCL's ring order is its own)
            tot = [torch.stack([parts[r][0][k] for r in range(self.world)]).sum(0)
                   for k in range(len(tensors))]
            self._sums[key] = [tot, self.world, self._mark(tot)]
        tot, left, ev = self._sums[key]
        self._wait(ev, tensors)
        for t, v in zip(tensors, tot):
            t.copy_(v)
            if v.is_cuda:  # the sum lives until the last reader's copy has run, on whatever stream
                v.record_stream(torch.cuda.current_stream(v.device))
        if left <= 1:
            del self._sums[key]
        else:
            self._sums[key][1] = left - 1


class _LocalComm:
    timing = False

    def __init__(self, world: LocalWorld, rank: int) -> None:
        self._w, self.rank, self.world = world, rank, world.world
        self._seq: dict = {}
        self._open: dict = {}

    def launch(self, tensors, tag=None) -> None:
        seq = self._seq.get(tag, 0)
        self._seq[tag] = seq + 1
        self._open[tag] = seq
        self._w._launch(self.rank, (tag, seq), tensors)

    def complete(self, tensors, tag=None) -> None:
        self._w._complete(self.rank, (tag, self._open.pop(tag)), tensors)

    def all_reduce_now(self, t, op=None) -> None:
        raise RuntimeError("LocalWorld has no blocking collectives: pass `rounds` / `item_counts` "
                           "computed over all ranks to the trainers instead")


class ItemSync:
    """Keeps the replicated tensors (item table, optional item bias) consistent across ranks.

    On a ROCm device the two elementwise passes around the all-reduce are the fused kernels
    ``bpr_item_delta`` / ``bpr_item_fold`` of libbprcore (one pass over the table each, buffers
    allocated once); on CPU tensors (gloo tests) the same algebra runs as torch ops.

    hot_rows > 0 (needs `engine`, whose item table must be tensors[0]): the hot tier — see the
    module docstring.  `item_counts` [I]: training positives per item over ALL ranks (the hot set
    must be the same everywhere); None = this rank's `local_items` are counted and all-reduced."""

    def __init__(self, tensors: list[torch.Tensor], group: Optional[dist.ProcessGroup] = None,
                 scale: float = 1.0, comm=None, engine=None, hot_rows: int = 0,
                 item_counts: Optional[torch.Tensor] = None,
                 local_items: Optional[torch.Tensor] = None, force_tiers: bool = False) -> None:
        self.tensors = [t.detach() for t in tensors if t is not None]
        self.group = group
        self.scale = scale  # 1.0: apply every rank's update; 1/world: DDP-style mean
        self.base = [t.clone() for t in self.tensors]
        self._own = [torch.empty_like(t) for t in self.tensors]
        self._tot = [torch.empty_like(t) for t in self.tensors]
        self._pending = False
        self._cuda = bool(self.tensors) and self.tensors[0].is_cuda
        self.comm = comm if comm is not None else _DistComm(group, self._cuda)
        self._lib = None
        if self._cuda:
            from revisit_bpr import native

            self._lib, self._check = native.load(), native.check
            for t in self.tensors:
                if t.dtype != torch.float32 or not t.is_contiguous():
                    raise ValueError("ItemSync needs contiguous float32 tensors")
        # ---- hot tier
        self.engine = engine
        self.hot_tier = False
        self.top_share = self.cold_top_share = 0.0
        self._hot_pending = False
        # force_tiers: run both tiers' passes on ONE rank (collectives are no-ops) — bench.py
        # --emulate-ranks measures what a rank's step costs beside its launch without a node
        if hot_rows > 0 and (self.world > 1 or force_tiers):
            self._setup_hot(int(hot_rows), item_counts, local_items)

    def _setup_hot(self, H: int, item_counts, local_items) -> None:
        e = self.engine
        if e is None or not self._cuda or e.Q.data_ptr() != self.tensors[0].data_ptr():
            raise ValueError("the hot tier needs the engine whose item table is tensors[0]")
        I = e.I
        if item_counts is None:
            if local_items is None:
                raise ValueError("hot tier: pass item_counts (all ranks) or this rank's local_items")
            item_counts = torch.bincount(local_items.long().reshape(-1), minlength=I).to(torch.int64)
            self.comm.all_reduce_now(item_counts)
        cnt = item_counts.detach().to("cpu", torch.int64).clone()
        cnt[0] = 0
        # share of the job's triples that update the most / the first not-hot popular row
        self.top_share = float(cnt.max()) / max(float(cnt.sum()), 1.0)
        H = min(H, int((cnt > 0).sum()))
        if H <= 0:
            return
        # the H most popular rows, ties by ascending id — the same list on every rank
        order = torch.argsort(cnt * (I + 1) + (I - torch.arange(I)), descending=True)[:H]
        e.set_hot_items(order.to(torch.int32), cnt.clamp(max=2**32 - 1))
        d = e.d
        self._hb = torch.empty(H, d, dtype=torch.float32, device=e.device)
        self._htot = torch.zeros(H, d, dtype=torch.float32, device=e.device)
        e.hot_tier_begin(self._hb)
        self.hot_tier = True
        self.hot_items = order
        rest = cnt.clone()
        rest[order] = 0
        self.cold_top_share = float(rest.max()) / max(float(cnt.sum()), 1.0)

    @property
    def timing(self) -> bool:
        return self.comm.timing

    @timing.setter
    def timing(self, on: bool) -> None:
        self.comm.timing = bool(on)

    @property
    def world(self) -> int:
        return self.comm.world

    @property
    def rank(self) -> int:
        return self.comm.rank

    def max_over_ranks(self, value: int) -> int:
        """MAX of a host integer over the group (e.g. chunks per epoch: ranks whose shard holds
        fewer chunks must still take part in every reconciliation, or the collectives of different
        ranks stop matching up)."""
        if self.world == 1:
            return int(value)
        dev = self.tensors[0].device if self.tensors else torch.device("cpu")
        t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
        self.comm.all_reduce_now(t, op=dist.ReduceOp.MAX)
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

    def timing_read(self) -> dict:
        """Average duration of the recorded all-reduces (ms, on the stream they ran on) per tier and
        their message sizes; clears the record."""
        torch.cuda.synchronize()
        ev = getattr(self.comm, "events", [])
        out = {}
        for tag in ("cold", "hot"):
            ms = [a.elapsed_time(b) for a, b, _, tg in ev if tg == tag]
            size = next((n for _, _, n, tg in ev if tg == tag), 0)
            out[tag] = {"all_reduces": len(ms), "all_reduce_ms_avg": (sum(ms) / len(ms)) if ms else None,
                        "message_bytes": size}
        if hasattr(self.comm, "events"):
            self.comm.events = []
        res = dict(out["cold"])
        res["hot"] = out["hot"]
        return res

    # ---- hot tier -------------------------------------------------------------------------------
    @torch.no_grad()
    def hot_step(self) -> None:
        """After every STREAM launch (or sub-launch) of the engine: fold the all-reduced hot block of
        the previous exchange, cut this launch's hot deltas and send them off.  Every rank calls it
        the same number of times (a rank whose launch was empty contributes zeros)."""
        if not self.hot_tier:
            return
        if self._hot_pending:
            self.comm.complete([self._htot], "hot")
        self.engine.hot_exchange(self._hb, self._htot, fold_prev=self._hot_pending, cut=True,
                                 cold_base=self.base[0])
        self.comm.launch([self._htot], "hot")
        self._hot_pending = True

    @torch.no_grad()
    def hot_finish(self) -> None:
        """Fold the hot exchange in flight (end of an epoch: the tables are read next)."""
        if not self.hot_tier or not self._hot_pending:
            return
        self.comm.complete([self._htot], "hot")
        self.engine.hot_exchange(self._hb, self._htot, fold_prev=True, cut=False,
                                 cold_base=self.base[0])
        self._hot_pending = False

    @property
    def can_fuse(self) -> bool:
        """`step_cut` is available: a ROCm engine whose item table is tensors[0]."""
        return self.engine is not None and self._lib is not None and bool(self.tensors) and \
            self.engine.Q.data_ptr() == self.tensors[0].data_ptr()

    @torch.no_grad()
    def step_cut(self) -> None:
        """`hot_step()` + `step()` + the cut of the engine's next adaptive snapshot as ONE pass over the
        item table (`bpr_sync_cut`), right after a `train_stream(cut=True)`: the three elementwise
        kernels a rank runs between two launches, and their boundaries, become one.  The same two
        all-reduces are launched afterwards, in the same order."""
        if self._hot_pending:
            self.comm.complete([self._htot], "hot")
        if self._pending:
            self.comm.complete(self._tot, "cold")
        hot = self.hot_tier
        self.engine.sync_cut(self._hb if hot else None, self._htot if hot else None, self._hot_pending,
                             self.base[0], self._own[0], self._tot[0], self.scale, 2 if self._pending else 1)
        for t, b, own, tot in list(zip(self.tensors, self.base, self._own, self._tot))[1:]:  # item bias
            if self._pending:
                self._check(self._lib.bpr_item_fold_delta(t.data_ptr(), b.data_ptr(), own.data_ptr(),
                                                          tot.data_ptr(), self.scale, t.numel(),
                                                          torch.cuda.current_stream().cuda_stream))
            else:
                self._delta(t, b, own, tot)
        if hot:
            self.comm.launch([self._htot], "hot")
            self._hot_pending = True
        self.comm.launch(self._tot, "cold")
        self._pending = True

    @torch.no_grad()
    def rebase(self) -> None:
        """The replicated tensors were overwritten IN PLACE (a checkpoint restored into the same
        storage, restore-best, STRICT steps between STREAM epochs): take them as the new reconciled
        state.  Exchanges in flight are completed and dropped, the cold bases are re-cut from the
        tensors, the hot base is re-gathered from the item table (without this the next `hot_step`
        would write Q[hot] = old base + deltas and silently revert those rows).  Every rank calls it,
        with the same tensors' content, at a point where the hot block holds no uncut deltas (after
        `hot_finish()` / a `hot_step()`: the end of an epoch)."""
        if self._hot_pending:
            self.comm.complete([self._htot], "hot")
            self._hot_pending = False
        if self._pending:
            self.comm.complete(self._tot, "cold")
            self._pending = False
        for t, b in zip(self.tensors, self.base):
            b.copy_(t)
        if self.hot_tier:
            self._htot.zero_()
            self.engine.hot_tier_begin(self._hb)

    def close(self) -> None:
        """Leave the hot tier (launches fold their hot block themselves again)."""
        if self.hot_tier:
            self.hot_finish()
            self.engine.hot_tier_end()
            self.hot_tier = False

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
        all-reduce on the side stream.  (Hot tier: call right after `hot_step` — the hot rows' cold
        base then equals their value and their cold delta is exactly zero.)"""
        if self._pending:
            self.finish()
        for t, b, own, tot in zip(self.tensors, self.base, self._own, self._tot):
            self._delta(t, b, own, tot)
        self.comm.launch(self._tot, "cold")
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
        self.comm.complete(self._tot, "cold")
        for t, b, own, tot in zip(self.tensors, self.base, self._own, self._tot):
            self._check(self._lib.bpr_item_fold_delta(t.data_ptr(), b.data_ptr(), own.data_ptr(),
                                                      tot.data_ptr(), self.scale, t.numel(),
                                                      torch.cuda.current_stream().cuda_stream))
        self.comm.launch(self._tot, "cold")

    @torch.no_grad()
    def finish(self, rebase: bool = False) -> None:
        """Fold the other ranks' contributions into the live replicas (one period late): the
        replica keeps what it learned since start(); the base becomes the reconciled cut.
        rebase=True (only valid when nothing trained since start()) sets replica = base."""
        if not self._pending:
            return
        self._pending = False
        self.comm.complete(self._tot, "cold")
        for t, b, own, tot in zip(self.tensors, self.base, self._own, self._tot):
            self._fold(t, b, own, tot, rebase)
