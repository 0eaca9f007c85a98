"""The throughput training loop: the reference's ``train_one_epoch`` (example.py:157-192) with the
whole inner loop on the device.

Per epoch: ``bpr_plan_epoch`` (seeded shuffle into chunks of one sampler-refresh period, grouped by
user) and then, per chunk, ``bpr_adaptive_refresh`` (adaptive sampler only) + ONE fused STREAM
launch (sample negative → gather → gradient → SGD scatter).  The host ships no per-batch data.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from revisit_bpr import engine as eng


def sort_cu_ms(I: int, d: int) -> float:
    """CU-milliseconds of one snapshot sort (one 1,024-thread workgroup per column, measured on MI355X): the
    binned sort of r5 (columns of 2,048 .. 20,480 keys: ~2.0 ns per key + 3 us per column — 42 us for ML-20M's
    20,108; up to 65,535 keys with G workgroups per column: MSD's 41,141 in 3 x 70 us) or the in-LDS radix sort
    (~5.05 ns per key, 1.3x when columns are split and merged: I > 36,864)."""
    if 2048 <= I <= 20480:
        return d * (2.0e-6 * I + 0.003)
    if I <= 65535:  # the binned sort with G workgroups per column, each reading the whole column (MSD: G = 3, 70 us each)
        G = -(-(I * 106 // 100) // (20 * 1024))
        return G * d * (1.6e-6 * I + 0.005)
    return 5.05e-6 * I * d * (1.3 if I > 36864 else 1.0)


def auto_refresh_cus(I: int, d: int, launch_triples: int, total_cus: int = 256) -> int:
    """CUs for the side stream of the overlapped snapshot schedule: enough that the sort of I x d
    keys finishes inside one STREAM launch, as few as possible because the launch loses them.
    Calibrated on MI355X (profiles/shapes_r03.txt, r05_binned_sort.md): `sort_cu_ms`, a launch ~0.72 / 0.8 / 1.1 /
    1.8 / 3.5 ns per triple at d <= 32 / 64 / 128 / 256 / 512; measured optima: 32 CUs for ML-20M d=128 since
    the binned sort (64 before it), 64 for MSD d=256 since the split binned sort (96 before it; r6 re-measured:
    462 M triples/s on 64 against 403 on 96)."""
    lag, cus = auto_schedule(I, d, launch_triples, total_cus)
    if lag > 0.0 and cus > 0:  # the split the cost model of the whole step picks
        return cus
    per_triple = 0.72e-6 if d <= 32 else 0.8e-6 if d <= 64 else 1.1e-6 if d <= 128 else \
        1.8e-6 if d <= 256 else 3.5e-6 * d / 512
    launch_ms = max(launch_triples, 1) * per_triple
    want = sort_cu_ms(I, d) / (0.95 * launch_ms)
    # multiples of 32 only: the driver deals the mask bits over 8 XCDs x 4 shader engines, and a
    # count that leaves the engines uneven costs the launch more than the CUs it frees (Yelp: 72 CUs
    # 570 M triples/s, 64 CUs 632 M)
    floor = 32 if 2048 <= I <= 20480 else 64
    return int(min(max(32 * math.ceil(want / 32 - 0.25), floor), total_cus // 2))


def auto_async_cut(I: int, refresh_cus: int) -> bool:
    """Should the transpose of the next snapshot's keys leave the launch stream (`bpr_train_stream_acut`)?  Only where
    the sorter has slack on its masked CUs — tables the one-workgroup binned sort serves (2,048 .. 20,480 items:
    ML-20M's sort takes 160 us of a 226-us launch on 32 CUs; +1.4 % steady, +1.7 % early).  MSD's split sort (833 us on
    64 CUs beside a 905-us launch) and Yelp's radix path have none: there the transpose beside the sort puts the sorter
    on the critical path (MSD early 508 -> 489 M, Yelp SGD 716 -> 430 M: profiles/r06_shapes.txt)."""
    return refresh_cus > 0 and 2048 <= I <= 20480


def lag_within_budget(lr: float, launch_triples: int, budget: Optional[float] = None) -> bool:
    """May the adaptive snapshot be one launch older (refresh_lag 1) — and may a CU keep the hottest rows in LDS
    for a launch (`hot_lds_rows`) — at this learning rate?  Both mean that a launch works on item state that misses
    up to two launches of updates, what a rank of a two-rank job at the rank cadence sees:
    lr x 2 x launch <= LAG_BUDGET.  Measured on the ML-20M-shaped set against the reference's OWN loop (lr 0.05:
    tests/golden/e2e_ml20m_reference_prefix.json) and against exact mini-batches, which that fixture pins
    (profiles/r05_fullepoch_reference.md, r06_parity_study.md; nDCG@100, 8 seeds): at lr 0.05 (19,917) the lagged
    schedule trails the reference by 0.0065 at the end of the first epoch; at lr 0.01 (3,983) it stays inside
    +-0.002 up to epoch 4 but LEADS exact mini-batches by +0.0022 / +0.0044 at epochs 6 / 8 (+0.0030 / +0.0056 with
    the LDS tier) — r5 had gated epochs 3-4 only and put the constant at 4,000, on that point; r6 moved it to
    2,000.  At the reference configs' lr 0.001 (398) lag 1 + LDS tier track exact mini-batches to +0.0006 / +0.0014 /
    +0.0016 / +0.0009 over the whole climb (40 ... 160 epochs)."""
    return lr * 2.0 * launch_triples <= (LAG_BUDGET if budget is None else budget)


def auto_schedule(I: int, d: int, launch_triples: int, total_cus: int = 256,
                  lr: Optional[float] = None) -> tuple[float, int]:
    """(refresh_lag, refresh_cus) for a shape (and, r5, a learning rate: `lag_within_budget` — outside
    the staleness budget the answer is the reference's schedule), from a two-line cost model calibrated on MI355X
    (profiles/shapes_r03.txt, shapes_r04.txt): the reference's schedule costs launch + sort on the
    whole chip; the overlapped one max(launch on the CUs it keeps, sort on the CUs it gets) — a sort
    that is only partly hidden still pays, a side stream that is too small does not (MSD d=256 with
    the sort on 64 CUs: 387 M triples/s against 393 M at lag 0; on 96 CUs the sort fits).  Returns
    lag 0 when no CU split beats the serial schedule."""
    if lr is not None and not lag_within_budget(lr, launch_triples):
        return 0.0, 0
    per_triple = 0.72e-6 if d <= 32 else 0.8e-6 if d <= 64 else 1.1e-6 if d <= 128 else \
        1.8e-6 if d <= 256 else 3.5e-6 * d / 512
    launch_ms = max(launch_triples, 1) * per_triple
    sort_ms_cu = sort_cu_ms(I, d)
    binned = 2048 <= I <= 20480
    serial = launch_ms + sort_ms_cu / (min(total_cus, d) if binned else total_cus) + 0.040  # three kernels and their boundaries
    best = (serial, 0.0, 0)
    for cus in (32, 64, 96, 128):
        if cus > total_cus // 2 or (cus == 32 and not binned):
            continue
        if binned:
            # 224 CUs run the launch as fast as 256 (profiles/r05_sort_upper_bound.md), +6 % per 64 CUs below;
            # one workgroup per column and CU: the sort takes whole rounds, 10 % slower beside a launch
            stretch = 1.0 + 0.06 * (cus - 32) / 64
            sort_ms = 1.1 * math.ceil(d / cus) * sort_ms_cu / d
        else:
            # the launch on the remaining CUs: +6 % per 64 CUs while the atomic units bound it
            # (d <= 128), more once HBM does (d >= 256: measured +3 % .. +5 % at 64)
            stretch = 1.0 + (0.06 if d <= 128 else 0.05) * cus / 64 * (1.0 if d <= 128 else 1.5)
            sort_ms = 0.9 * sort_ms_cu / cus
        step = max(launch_ms * stretch, sort_ms) + 0.026  # cut + the launch-to-launch gap
        if step < best[0]:
            best = (step, 1.0, cus)
    return best[1], best[2]


# Staleness budget of a multi-rank job (profiles/r04_cadence_study.txt).  A rank sees the
# other ranks' item updates — and re-sorts its sampler snapshot — once per chunk, folded one chunk
# late: up to 2 x world x chunk triples of the job are applied against rows and a snapshot that do
# not know them yet.  The study (full ML-20M shape, 1/2/4/8 ranks, three learning rates) puts the
# edge of the +-0.002 nDCG@100 band at  lr x world x chunk ~ 4,000  (lr is per triple: the loss is a
# sum): at lr 0.05 that is period / (2.5 world) — less than the single-GPU cadence counted in job
# triples, which does miss the band at 8 ranks — and at the reference configs' lr 0.001 a full period
# per rank fits with room to spare.
STALENESS_BUDGET = 4_000.0
# ... and of ONE rank's own shortcuts (a snapshot one launch older, hot rows one launch stale per CU): r6 — see
# `lag_within_budget`
LAG_BUDGET = 2_000.0


MAX_CHUNKS_PER_RANK_SHARE = 4  # chunks may shrink to period / (4 x world), no further

# LDS tier of the hot block (r6, `bpr_set_hot_lds`): rows asked for — the library keeps as many as fit a CU's
# LDS beside the seen bitmaps (ML-20M, d = 128: 128) and takes the tier only for launches that fill the chip
HOT_LDS_ROWS = 512


def hot_lds_rows(lr: float, launch_triples: int, world: int = 1, budget: Optional[float] = None) -> int:
    """Rows of the hot block a CU may keep in LDS for a launch of this size at this learning rate (0: none).
    With the tier a workgroup sees the OTHER workgroups' updates of those rows one launch late.  One rank: that is
    the staleness of the lagged snapshot, so the same rule — lr x 2 x launch <= LAG_BUDGET (inside at the reference
    configs' lr 0.001 and up to 0.005 for an ML-20M period, outside at 0.01 and 0.05;
    tests/test_gpu_fullscale_reference.py gates both sides).  Several ranks: a rank never sees the other ranks'
    updates of a launch anyway ((N - 1) / N of them; the hot tier exchanges them after the launch); the tier makes
    that (N - 1/256) / N — priced as one more rank in the cadence's own budget: lr x (N + 1) x launch <=
    STALENESS_BUDGET (reasoned from the r4 study, not measured with the tier on)."""
    if max(world, 1) > 1:
        return HOT_LDS_ROWS if lr * (world + 1) * launch_triples <= (STALENESS_BUDGET if budget is None else budget) else 0
    return HOT_LDS_ROWS if lag_within_budget(lr, launch_triples, budget) else 0


def launches_per_period(lr: float, world: int, period: int, budget: Optional[float] = None) -> int:
    """Chunks a rank cuts a refresh period into so that lr x world x chunk stays inside the budget:
    1 (a full period per rank: "rank" cadence) .. world (period / world: "job" cadence, the
    single-GPU cadence counted in job triples) .. 4 x world (aggressive learning rates: lr 0.05 at
    8 ranks misses the band at the job cadence, profiles/r04_cadence_study.txt)."""
    if world <= 1:
        return 1
    budget = STALENESS_BUDGET if budget is None else budget
    return int(min(max(math.ceil(period * lr * world / budget), 1), MAX_CHUNKS_PER_RANK_SHARE * world))


class StreamTrainer:
    def __init__(self, model, users: torch.Tensor, items: torch.Tensor, seen_indptr: torch.Tensor,
                 seen_indices: torch.Tensor, lr: float, sampler: str = "adaptive",
                 adaptive_p: float = 0.01, batch_size: int = 256, seed: int = 13,
                 max_inflight: Optional[int] = None, run_len: int = 0, rank: int = 0,
                 item_sync=None, sync_every: int = 1, world: Optional[int] = None,
                 refresh_lag: float | str = 0.0, refresh_split: int = 1, refresh_cus: int = 0,
                 shard_refresh: bool = False, cadence: str = "job", hot_split: int = 1,
                 rounds: Optional[int] = None, jit_plan: bool = False, async_cut: bool | str = "auto",
                 hot_lds: int | str = "auto", launch_split: int | str = "auto") -> None:
        """model: revisit_bpr.models.BPR on a ROCm device; users/items: int32 training triples on
        the device; seen CSR: int64 indptr [U+1], int32 indices.  `batch_size` only sets the
        adaptive refresh period int(I·ln I / batch_size) batches, as example.py:302.

        Snapshot schedule of the adaptive sampler (extensions; the defaults are the reference's
        ``update_stats`` every period, neg_samplers.py:122-132):
          refresh_split k   the period is cut into k launches and the snapshot retaken before each;
          refresh_lag       "auto": 1 with the sort on masked CUs when the shape gains from it AND the learning
                            rate keeps the older snapshot inside the staleness budget (`auto_schedule`,
                            `lag_within_budget`), else 0.
                            0: the snapshot is sorted between launches (the launch waits for it).
                            1: the snapshot a launch reads was cut BEFORE the previous launch and
                               sorted beside it (`adaptive_refresh_begin` / `_commit`): its age runs
                               from one to two launches instead of zero to one, nothing waits.
                            0 < f < 1: a launch is cut at 1 - f; the next launch's snapshot is taken
                               there and sorted beside the remainder.
          refresh_cus n     n > 0: the sort runs on a stream masked to n of the chip's CUs and the
                            STREAM kernel on the complementary mask (0: unmasked side stream;
                            -1: chosen from the shape, `auto_refresh_cus`).
          shard_refresh     several ranks (item_sync): every rank sorts d / world factors and an
                            all-gather shares the orders (Engine.adaptive_refresh_sharded) instead
                            of every rank sorting all of them; refresh_lag 0 only.

        Several ranks (item_sync), r4:
          cadence           "job": a chunk is 1 / world of the refresh period, so the snapshot and the
                            item reconciliation keep the single-GPU cadence counted in triples of
                            the whole job (launches shrink with the number of ranks).
                            "rank": every rank launches a FULL period; refresh and cold reconciliation
                            once per rank-period, the hot rows (item_sync's hot tier) after every
                            launch.
                            "auto": between the two — the largest chunk that keeps
                            lr x world x chunk inside STALENESS_BUDGET (`launches_per_period`).
          hot_split k       a chunk runs as k launches with a hot-tier exchange after each.
          rounds            chunks per epoch over all ranks (None: a MAX all-reduce decides).

        jit_plan (refresh_lag = 1 only): the epoch is never planned as a whole — chunk k + 1 is planned
        by `bpr_plan_chunk` on the side stream behind the sort of chunk k (the plan does not depend on
        the model; the sorter idles ~40 us per step), and `adaptive_refresh_commit` waits for both.
        Same chunks as `bpr_plan_epoch` makes (same members, grouped by user).

        launch_split k (refresh_lag 0, one GPU): a refresh period runs as k launches that read the SAME snapshot —
        a launch walks its triples grouped by user, so a user's triples of a period are otherwise applied back to
        back; k launches deal them into k groups placed apart, as the reference's shuffled mini-batches do.  "auto"
        (default): 2 outside the one-rank budget (high learning rates: at the end of the first lr-0.05 epoch at the
        ML-20M shape STREAM reads -0.0013 / -0.0018 nDCG@100 / Recall@20 against the reference's own loop with one
        launch per period, -0.0012 / -0.0012 with two, 12 seeds each: profiles/r06_parity_study.md), else 1.
        With the uniform sampler (no snapshot to share) it only halves the launches.

        hot_lds: rows of the hot block a CU keeps in LDS during a launch (`bpr_set_hot_lds`; r6): "auto" =
        `hot_lds_rows` — on inside the staleness budget, off outside; 0 = off; n = asked for whatever the rate.

        async_cut (refresh_lag = 1, one GPU): the transpose of the next snapshot's keys leaves the launch stream —
        a read-only pass on the side stream beside the NEXT launch (`bpr_train_stream_acut`; r6: the fold of the
        hot block stays on the launch stream, so the LDS tier stays in use): +1.4 % on the metric's configuration,
        the same curves (profiles/r06_parity_study.md).  "auto" (default): `auto_async_cut` — on with a masked side
        stream where the sorter has slack (tables of 2,048 .. 20,480 items)."""
        if users.dtype != torch.int32 or items.dtype != torch.int32:
            raise ValueError("users / items must be int32 device tensors")
        auto_lag = isinstance(refresh_lag, str)
        if auto_lag and refresh_lag != "auto":
            raise ValueError("refresh_lag must be in [0, 1] or 'auto'")
        if not auto_lag and not 0.0 <= refresh_lag <= 1.0 or refresh_split < 1:
            raise ValueError("refresh_lag must be in [0, 1], refresh_split >= 1")
        self.model = model
        self.engine = model.engine()
        self.users, self.items = users.contiguous(), items.contiguous()
        self.n = users.numel()
        self._users_sorted = eng.Engine.users_sorted(self.users)  # (CSR order: the plan takes one radix pass)
        self.engine.bind_seen_csr(seen_indptr, seen_indices)
        model._has_csr = True
        self.engine.set_optimizer(eng.OPT_SGD, lr=lr)
        self.sampler = {"adaptive": eng.NEG_ADAPTIVE, "uniform": eng.NEG_UNIFORM}[sampler]
        self.adaptive_p = adaptive_p
        I = self.engine.I
        every = max(1, int(I * math.log(I) / batch_size))
        # one refresh period = every*batch_size triples of the WHOLE job: with the users sharded
        # over `world` ranks each rank advances 1/world of it per chunk, so the snapshot refresh
        # and the item reconciliation keep their single-GPU cadence
        if world is None:
            world = item_sync.world if item_sync is not None else 1
        if cadence not in ("job", "rank", "auto"):
            raise ValueError("cadence must be 'job', 'rank' or 'auto'")
        self.cadence = cadence
        per_period = (max(world, 1) if cadence == "job" else 1 if cadence == "rank" else
                      launches_per_period(lr, max(world, 1), every * batch_size, STALENESS_BUDGET))
        if isinstance(launch_split, str):
            if launch_split != "auto":
                raise ValueError("launch_split must be an int >= 1 or 'auto'")
            launch_split = 1 if (max(world, 1) > 1 or refresh_split != 1 or (not auto_lag and refresh_lag != 0.0)
                                 or lag_within_budget(lr, every * batch_size)) else 2
        self.launch_split = max(1, int(launch_split))
        if self.launch_split > 1 and (item_sync is not None or refresh_split != 1):
            raise ValueError("launch_split > 1: one GPU, refresh_split 1")
        self.chunk = max(1, min(every * batch_size // (per_period * refresh_split * self.launch_split), self.n))
        self.hot_split = max(1, int(hot_split))
        self.hot_lds = hot_lds_rows(lr, self.chunk, max(world, 1)) if isinstance(hot_lds, str) else int(hot_lds)
        self.engine.set_hot_lds(self.hot_lds)
        U = self.engine.U
        # staleness budget: at most ~U/4 triples in flight against one parameter cut
        self.max_inflight = max(64, U // 4) if max_inflight is None else max_inflight
        self.engine.set_stream_opts(True, run_len)
        if auto_lag:  # by shape and learning rate; one GPU (several ranks keep their cadence's schedule)
            refresh_lag, refresh_cus = (auto_schedule(I, self.engine.d, self.chunk, lr=lr)
                                        if item_sync is None and self.sampler == eng.NEG_ADAPTIVE else (0.0, 0))
        elif refresh_lag >= 1.0 and self.sampler == eng.NEG_ADAPTIVE and item_sync is None and \
                not lag_within_budget(lr, self.chunk):
            import warnings

            warnings.warn(f"refresh_lag 1 at lr {lr} with launches of {self.chunk} triples is outside the staleness "
                          f"budget (lr x 2 x launch = {lr * 2 * self.chunk:.0f} > {LAG_BUDGET:.0f}): the "
                          "rising part of the curve leaves the reference's by more than 0.002 nDCG@100 "
                          "(profiles/r05_fullepoch_reference.md, r06_parity_study.md); refresh_lag='auto' picks by learning rate",
                          stacklevel=2)
        self.refresh_lag = float(refresh_lag) if self.sampler == eng.NEG_ADAPTIVE else 0.0
        if self.refresh_lag != 0.0 and self.launch_split > 1:
            raise ValueError("launch_split > 1 needs refresh_lag 0")
        self.shard_refresh = bool(shard_refresh) and item_sync is not None and item_sync.world > 1
        if self.shard_refresh and self.refresh_lag != 0.0:
            raise ValueError("shard_refresh needs refresh_lag = 0")
        self._main = self._side = None
        if self.refresh_lag > 0.0 and refresh_cus != 0:
            total = torch.cuda.get_device_properties(users.device).multi_processor_count
            if refresh_cus < 0:
                refresh_cus = auto_refresh_cus(I, self.engine.d, self.chunk, total)
            self._side = eng.MaskedStream(users.device, eng.cu_mask(0, refresh_cus, total))
            self._main = eng.MaskedStream(users.device,
                                          eng.cu_mask(refresh_cus, total - refresh_cus, total))
            self.engine.set_side_stream(self._side)
        self.seed, self.rank = seed, rank
        self.epoch = 0
        self.drawn = 0
        self.jit_plan = bool(jit_plan) and self.refresh_lag >= 1.0
        if self.jit_plan:  # two chunk buffers: launch k reads one while chunk k + 1 is planned into the other
            self._cb = [(torch.empty(self.chunk, dtype=torch.int32, device=users.device),
                         torch.empty(self.chunk, dtype=torch.int32, device=users.device)) for _ in range(2)]
            self._pu = self._pi = None
            self._gk, self._planned = 0, None
        else:
            self._pu = torch.empty_like(self.users)
            self._pi = torch.empty_like(self.items)
        self._scalars = torch.zeros(4, dtype=torch.float32, device=users.device)
        self._synced = False  # this chunk's reconciliation already ran (fused into the launch's cut)
        if isinstance(async_cut, str):
            if async_cut != "auto":
                raise ValueError("async_cut must be a bool or 'auto'")
            async_cut = self._side is not None and auto_async_cut(I, refresh_cus)
        self.async_cut = bool(async_cut) and self.refresh_lag >= 1.0 and item_sync is None
        self.item_sync, self.sync_every = item_sync, sync_every
        # shards are balanced by interactions, not equal: every rank runs the same number of
        # rounds per epoch (a rank out of triples still joins the item reconciliations)
        self.rounds = -(-self.n // self.chunk)
        if rounds is not None:
            self.rounds = int(rounds)
        elif item_sync is not None:
            self.rounds = item_sync.max_over_ranks(self.rounds)

    # The epoch is written as a generator that yields after every launch (+ hot-tier exchange) and
    # after every chunk: `train_epoch` just runs it; a LocalWorld simulation (several ranks in one
    # process, revisit_bpr.distributed) resumes the ranks' generators round-robin.
    def _launch(self, lo: int, hi: int, cut: bool = False, base: int = 0):
        hot = self.item_sync is not None and self.item_sync.hot_tier
        k = self.hot_split if hot else 1
        pu, pi = self._cb[self._gk & 1] if self.jit_plan else (self._pu, self._pi)
        lo, hi = lo - base, hi - base  # (jit_plan: positions inside the chunk buffer)
        for p in range(k):  # every rank runs k pieces, empty ones included: the exchanges line up
            a, b = lo + (hi - lo) * p // k, lo + (hi - lo) * (p + 1) // k
            if b > a:
                self.engine.train_stream(pu[a:b], pi[a:b], sampler=self.sampler,
                                         adaptive_p=self.adaptive_p, seed=self.seed,
                                         offset=(self.rank << 40) + self.drawn,
                                         max_inflight=self.max_inflight, scalars=self._scalars,
                                         cut=("async" if self.async_cut else True) if (cut and p == k - 1) else False)
                self.drawn += b - a
                if cut and p == k - 1 and self.item_sync is not None:
                    # several ranks: the launch left its epilogue to ONE pass that also runs the
                    # hot-tier step, the cold step and the cut of the next snapshot (bpr_sync_cut)
                    self.item_sync.step_cut()
                    self._synced = True
                    yield
                    continue
            if hot:
                self.item_sync.hot_step()
            yield

    def _plan(self, epoch: int, index: int, slot: int, on_side: bool) -> None:
        self.engine.plan_chunk(self.users, self.items, self.chunk, self.seed + epoch, index,
                               out=self._cb[slot & 1], on_side=on_side)

    def _chunk(self, lo: int, hi: int):
        e, lag = self.engine, self.refresh_lag
        if self.sampler != eng.NEG_ADAPTIVE:
            yield from self._launch(lo, hi)
            return
        if lag == 0.0:
            if self.shard_refresh:
                e.adaptive_refresh_sharded(self.item_sync.rank, self.item_sync.world, self.item_sync.group)
            elif (lo // self.chunk) % self.launch_split == 0:  # (launch_split: the period's later launches reuse it)
                e.adaptive_refresh()
            yield from self._launch(lo, hi)
            return
        # with one launch per snapshot (lag 1) and nothing touching the item table between two
        # launches (no item reconciliation) the keys of the NEXT snapshot are cut by the epilogue
        # of this launch: `begin` then only queues the sort
        # ... and with an item reconciliation: when that reconciliation and the cut are one pass
        # (ItemSync.step_cut after the launch: one reconciliation per chunk, engine-backed)
        fused = lag >= 1.0 and (self.item_sync is None or
                                (self.item_sync.can_fuse and self.sync_every == 1 and self.item_sync.hot_tier))
        if e.refresh_pending():
            e.adaptive_refresh_commit()  # the snapshot cut before the previous launch
        else:
            e.adaptive_refresh()         # first launch: nothing in flight yet
        cut = lo if lag >= 1.0 else min(hi, lo + max(1, int(round((1.0 - lag) * (hi - lo)))))
        # (fractional lag: BOTH phases always run — a short last chunk whose cut collapses onto its end
        # still owes the other ranks the hot-tier exchanges of an empty second phase)
        if cut > lo:
            yield from self._launch(lo, cut)
        base = 0
        if self.jit_plan:
            index = lo // self.chunk
            if self._planned != (self.epoch, index):  # the first chunk: on the launch stream
                self._plan(self.epoch, index, self._gk, False)
            base = lo
        e.adaptive_refresh_begin()
        if self.jit_plan:  # the next chunk, behind the sort just queued
            nxt = (self.epoch, index + 1) if (index + 1) * self.chunk < self.n else (self.epoch + 1, 0)
            self._plan(nxt[0], nxt[1], self._gk + 1, True)
            self._planned = nxt
        if cut < hi or lag < 1.0:
            yield from self._launch(cut, hi, cut=fused, base=base)
        if self.jit_plan:
            self._gk += 1

    def stream_scope(self):
        """Context in which this trainer's calls must run: its CU-masked launch stream, if any."""
        import contextlib

        return torch.cuda.stream(self._main.torch) if self._main is not None else contextlib.nullcontext()

    def train_epoch(self) -> dict:
        self.epoch_begin()
        with self.stream_scope():  # the whole epoch on the CU-masked stream
            for _ in self.epoch_iter():
                pass
        return self.epoch_end()

    def train_chunks(self, n: int) -> dict:
        """The next `n` chunks (refresh periods) of the epoch in progress — a new epoch is begun, i.e.
        planned, when none is; an epoch that ends on the way is closed and the call returns.  For a
        training PREFIX (tests/test_gpu_fullscale_reference.py: the reference's own loop is timed in
        refresh periods there); the statistics are those of the epoch so far."""
        self.epoch_begin()  # the launch stream waits for whatever read the tables since (an evaluation)
        if getattr(self, "_it", None) is None:
            self._it = self.epoch_iter()
        done = 0
        with self.stream_scope():
            while done < n:
                try:
                    done += next(self._it) == "chunk"
                except StopIteration:
                    self._it = None
                    break
        return self.epoch_end()

    def epoch_begin(self) -> None:
        if self._main is not None:
            self._main.torch.wait_stream(torch.cuda.current_stream(self.users.device))
        # from here to epoch_end nobody but the launches writes the item_bias: they may keep their
        # one-item-per-line copy of it instead of re-reading the vector every launch — unless an item
        # reconciliation carries the bias too (several ranks): ItemSync folds the other ranks' bias deltas into
        # the vector BETWEEN launches, through entry points that know no ctx, so every launch must refill
        # (with tracking on, the next launch's epilogue wrote its stale copy back over the
        # reconciled vector and the replicas drifted)
        sync_writes_bias = self.item_sync is not None and self.engine.item_bias is not None and any(
            t.data_ptr() == self.engine.item_bias.data_ptr() for t in self.item_sync.tensors)
        self.engine.set_bias_tracking(not sync_writes_bias)

    def epoch_end(self) -> dict:
        self.engine.set_bias_tracking(False)
        if self._main is not None:
            torch.cuda.current_stream(self.users.device).wait_stream(self._main.torch)
        if self.async_cut and self.engine.tuning("acut_fold", 1) == 0:
            # r4's form of the asynchronous cut: fold what the cuts left, wait for the sums they deliver on the side
            # stream (r6's default keeps fold and sums on the launch stream: nothing to do)
            self.engine.hot_fold()
            torch.cuda.synchronize(self.users.device)
        sc = self._scalars.tolist()
        cnt = max(sc[3], 1.0)
        return {"bpr_loss": sc[0] / cnt, "l2_reg": sc[1] / cnt, "logits_diff": sc[2] / cnt,
                "loss": (sc[0] + sc[1]) / cnt, "triples": int(sc[3])}

    def epoch_iter(self):
        """One epoch as a generator (see `_launch`); resume it inside `stream_scope()`."""
        e = self.engine
        self.model._reset_reg()
        if not self.jit_plan:
            e.plan_epoch(self.users, self.items, self.chunk, self.seed + self.epoch,
                         out=(self._pu, self._pi), sorted_input=self._users_sorted)
        self._scalars.zero_()
        hot = self.item_sync is not None and self.item_sync.hot_tier
        for k in range(self.rounds):
            lo = k * self.chunk
            hi = min(lo + self.chunk, self.n)
            if lo < hi:
                yield from self._chunk(lo, hi)
            else:
                # a rank that ran out of triples (shards are balanced by interactions, not equal)
                # still owes the others every collective of the round, in the same order: its share
                # of a sharded sort + all-gather, and (hot tier) one exchange per launch
                if self.shard_refresh and self.sampler == eng.NEG_ADAPTIVE:
                    e.adaptive_refresh_sharded(self.item_sync.rank, self.item_sync.world,
                                               self.item_sync.group)
                launches = 2 if 0.0 < self.refresh_lag < 1.0 else 1
                for _ in range(launches * (self.hot_split if hot else 1)):
                    if hot:
                        self.item_sync.hot_step()
                    yield
            if self.item_sync is not None and (k + 1) % self.sync_every == 0 and not self._synced:
                self.item_sync.step()
            self._synced = False
            yield "chunk"
        if self.item_sync is not None:
            self.item_sync.hot_finish()
            self.item_sync.finish()
        self.epoch += 1


class StrictTrainer:
    """Epochs of the reference's exact mini-batch loop (sample → model(batch) → backward →
    optimizer.step()) inside the library, for every optimizer the engine mirrors (SGD, momentum /
    Nesterov, Adam, RMSprop — the dense torch.optim semantics, see Model.train_strict).

    One GPU: one `bpr_train_strict` call per epoch.  Several: every rank runs the mini-batches of
    its own user shard and the replicated item table is reconciled by `item_sync` every
    refresh period / world batches (the STREAM cadence); optimizer state of the item table stays
    local to the rank, rows are brought to "now" (lazy replay flushed) before every reconciliation."""

    def __init__(self, model, optimizer, users: torch.Tensor, items: torch.Tensor,
                 seen_indptr: torch.Tensor, seen_indices: torch.Tensor, sampler: str = "adaptive",
                 adaptive_p: float = 0.01, batch_size: int = 256, seed: int = 13, rank: int = 0,
                 item_sync=None, world: Optional[int] = None,
                 order_seed: Optional[int] = None) -> None:
        """order_seed: the epoch permutations come from ``np.random.default_rng(order_seed)`` (one
        ``permutation(n)`` per epoch, the DataLoader(shuffle=True) stand-in of
        tests/golden/make_golden_e2e.py) instead of a device randperm seeded by `seed` — parity
        runs feed the reference's own epoch order."""
        if users.dtype != torch.int32 or items.dtype != torch.int32:
            raise ValueError("users / items must be int32 device tensors")
        self.model, self.optimizer = model, optimizer
        self.engine = model.engine()
        self.users, self.items = users.contiguous(), items.contiguous()
        self.n = users.numel()
        model.bind_seen_csr(seen_indptr, seen_indices)
        self.sampler = {"adaptive": eng.NEG_ADAPTIVE, "uniform": eng.NEG_UNIFORM}[sampler]
        self.adaptive_p, self.batch = adaptive_p, batch_size
        I = self.engine.I
        self.every = max(1, int(I * math.log(I) / batch_size))
        if world is None:
            world = item_sync.world if item_sync is not None else 1
        self.item_sync = item_sync
        self.chunk_batches = max(1, self.every // max(world, 1))
        self.seed, self.rank = seed, rank
        self.gen = torch.Generator(device=users.device).manual_seed(seed * 1000003 + rank)
        self._order_rng = None if order_seed is None else np.random.default_rng(order_seed)
        self.drawn = 0
        self._scalars = torch.zeros(4, dtype=torch.float32, device=users.device)
        self._fresh = False
        self.rounds = -(-self.n // (self.chunk_batches * batch_size))
        if item_sync is not None:  # same number of reconciliations on every rank (see StreamTrainer)
            self.rounds = item_sync.max_over_ranks(self.rounds)

    def train_epoch(self) -> dict:
        model, e, B = self.model, self.engine, self.batch
        model.train()
        if self._order_rng is not None:
            perm = torch.from_numpy(self._order_rng.permutation(self.n)).to(self.users.device)
        else:
            perm = torch.randperm(self.n, device=self.users.device, generator=self.gen)
        u, i = self.users[perm].contiguous(), self.items[perm].contiguous()
        adaptive = self.sampler == eng.NEG_ADAPTIVE
        self._scalars.zero_()
        if adaptive and not self._fresh:
            e.adaptive_refresh()
            self._fresh = True
        if self.item_sync is None:
            model.train_strict(self.optimizer, u, i, B, self.sampler, adaptive_p=self.adaptive_p,
                               seed=self.seed, offset=(self.rank << 40) + self.drawn,
                               refresh_every=self.every if adaptive else 0, scalars=self._scalars)
        else:
            step = self.chunk_batches * B
            for k in range(self.rounds):
                lo = k * step
                hi = min(lo + step, self.n)
                if lo < hi:
                    model.train_strict(self.optimizer, u[lo:hi], i[lo:hi], B, self.sampler,
                                       adaptive_p=self.adaptive_p, seed=self.seed,
                                       offset=(self.rank << 40) + self.drawn + lo, refresh_every=0,
                                       scalars=self._scalars)
                e.flush_items()
                self.item_sync.step()
                if adaptive:
                    e.adaptive_refresh()
            self.item_sync.finish()
        self.drawn += self.n
        sc = self._scalars.tolist()
        cnt = max(sc[3], 1.0)
        return {"bpr_loss": sc[0] / cnt, "l2_reg": sc[1] / cnt, "logits_diff": sc[2] / cnt,
                "loss": (sc[0] + sc[1]) / cnt, "triples": int(sc[3])}


class BatchedStreamTrainer:
    """Epochs of the reference's mini-batch loop with ANY torch.optim optimizer the engine mirrors
    (SGD, momentum / Nesterov, Adam, RMSprop) as fused launches: per epoch a seeded device shuffle
    (`bpr_shuffle_epoch`), then per sampler-refresh period `bpr_adaptive_refresh` + ONE
    `bpr_train_stream_batched` launch whose virtual mini-batches of `batch_size` triples take one
    dense optimizer step each (lazy replay for the rows a batch does not touch).  This is the
    throughput path of the Adam configs (configs/RQ3/time-split/ada-sampling-adam.yaml.j2);
    `StrictTrainer` is the same loop with a launch triple per batch and no asynchrony.

    Several GPUs: users sharded, item table (+ bias) reconciled by `item_sync` every refresh period
    of the whole job, optimizer state of the item table local to the rank (as StrictTrainer)."""

    def __init__(self, model, optimizer, users: torch.Tensor, items: torch.Tensor,
                 seen_indptr: torch.Tensor, seen_indices: torch.Tensor, sampler: str = "adaptive",
                 adaptive_p: float = 0.01, batch_size: int = 256, seed: int = 13,
                 max_inflight: Optional[int] = None, rank: int = 0, item_sync=None,
                 world: Optional[int] = None) -> None:
        if users.dtype != torch.int32 or items.dtype != torch.int32:
            raise ValueError("users / items must be int32 device tensors")
        self.model, self.optimizer = model, optimizer
        self.engine = model.engine()
        self.users, self.items = users.contiguous(), items.contiguous()
        self.n = users.numel()
        model.bind_seen_csr(seen_indptr, seen_indices)
        self.sampler = {"adaptive": eng.NEG_ADAPTIVE, "uniform": eng.NEG_UNIFORM}[sampler]
        self.adaptive_p, self.batch = adaptive_p, batch_size
        I, U = self.engine.I, self.engine.U
        every = max(1, int(I * math.log(I) / batch_size))  # example.py:302
        if world is None:
            world = item_sync.world if item_sync is not None else 1
        self.chunk = max(batch_size, every // max(world, 1) * batch_size)
        # staleness budget: at most ~U/4 triples against one parameter cut — and at most
        # I/3 (r5): every triple in flight holds two item rows, so beyond I/2 of them a row is, on average,
        # in two virtual batches at once; gradients of neighbouring batches that reach a row before either
        # step is closed merge into ONE optimizer step, which Adam normalises as one — on a 1,500-item table
        # with 1,000 triples in flight that reads -0.0020 nDCG@100 / -0.0024 Recall@20 at epoch 12 against
        # the reference over epoch orders, with 500 in flight -0.0014 / -0.0017, with one batch -0.0011 /
        # -0.0013 (100 seeds each, profiles/r05_vstream_inflight.txt).  No BASELINE shape but ML-20M
        # (6,702 against the chip's ~8,192 resident triples) is touched by the item bound.
        self.max_inflight = max(64, min(U // 4, I // 3)) if max_inflight is None else max_inflight
        self.seed, self.rank = seed, rank
        self.epoch = 0
        self.drawn = 0
        self.item_sync = item_sync
        self._pu = torch.empty_like(self.users)
        self._pi = torch.empty_like(self.items)
        self._scalars = torch.zeros(4, dtype=torch.float32, device=users.device)
        self.rounds = -(-self.n // self.chunk)
        if item_sync is not None:
            self.rounds = item_sync.max_over_ranks(self.rounds)

    def train_epoch(self) -> dict:
        model, e = self.model, self.engine
        model.train()
        e.shuffle_epoch(self.users, self.items, (self.seed << 20) + (self.rank << 12) + self.epoch,
                        out=(self._pu, self._pi))
        self._scalars.zero_()
        adaptive = self.sampler == eng.NEG_ADAPTIVE
        for k in range(self.rounds):
            lo = k * self.chunk
            hi = min(lo + self.chunk, self.n)
            if lo < hi:
                if adaptive:
                    e.adaptive_refresh()  # brings the item rows to "now" first
                model.train_stream_batched(self.optimizer, self._pu[lo:hi], self._pi[lo:hi],
                                           self.batch, self.sampler, adaptive_p=self.adaptive_p,
                                           seed=self.seed, offset=(self.rank << 40) + self.drawn,
                                           max_inflight=self.max_inflight, scalars=self._scalars)
                self.drawn += hi - lo
            if self.item_sync is not None:
                e.flush_items()
                self.item_sync.step()
        if self.item_sync is not None:
            self.item_sync.finish()
        self.epoch += 1
        sc = self._scalars.tolist()
        cnt = max(sc[3], 1.0)
        return {"bpr_loss": sc[0] / cnt, "l2_reg": sc[1] / cnt, "logits_diff": sc[2] / cnt,
                "loss": (sc[0] + sc[1]) / cnt, "triples": int(sc[3])}
