#!/usr/bin/env python
"""bench.py — BPR triples/s on MI355X for the configuration BASELINE.json's metric is quoted on.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (config.workload): ML-20M-shaped synthetic interactions (136,677 users x 20,108 items,
~9.6 M training triples after the user-split hold-out), d = 128, plain SGD + L2, ADAPTIVE negative
sampling p = 1/100 — BASELINE.json configs[2], the d=128 config the metric names.

One step = one refresh period of the reference's training loop (example.py:172-180 +
neg_samplers.py:122-123): `bpr_adaptive_refresh` (AdaptiveSampler.update_stats) followed by ONE
fused STREAM launch over the next int(I·ln I / 256)·256 shuffled triples (sample negative → gather
→ gradient → SGD scatter), all inputs resident in HBM.  Multi-GPU: weak scaling — every rank owns an
ML-20M-shaped user shard; the item table is replicated and reconciled by an asynchronous
all-reduce of item deltas every --sync-every steps (revisit_bpr/distributed.py).

Prints ONE JSON line (rank 0).  `value` (r6) = the TRAINED state: after the driver's K-step region (`timed_region`)
and three whole epochs from random init (`early_state` / `sustained`: what r1-r5 called `value`) the job trains on,
untimed, to 30 epochs, and epochs 31..130 are timed by wall clock between barriers, every `bpr_plan_epoch` in place,
nothing modelled (`steady_state`; its `first_epoch` is epoch 31 alone) — once the adaptive sampler has a model to
adapt to its negatives concentrate on popular rows and a launch costs more than on tables fresh from their random
init: a long job lives in that state.  `steps` = the steps behind `value`.  `roofline`: dominant kernel = the fused
stream kernel on that state, HBM bound, algorithmic bytes 24·d+8 per triple, duration from hipEvents recorded by
the library on the launch stream; `traffic` REPLAYED from committed rocprofv3 --pmc passes of the same state
(profiles/traffic_r06.json), never measured in-run.  `cpu_baseline`: the reference's op sequence restated on CPU
tensors, `cpu_baseline_c_port`: the C oracle with OpenMP — both on this host's cores over a bounded sample.
The schedule (snapshot one launch older on 32 masked CUs, asynchronous cut, LDS tier of the hot block) is what
`fast.StreamTrainer` picks by itself at the workload's learning rate (`fast.lag_within_budget`).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "revisit-bpr_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is achievable
# full-line fp32 atomic requests the chip retires per second (profiles/ubench_atomic_r02.txt:
# tools/ubench/atomic_bench.hip, uniform item popularity) — what actually bounds k_stream
ATOMIC_LINES_PEAK = 9.5e9
# default snapshot schedule of the adaptive sampler: the one the parity gates hold
# (tests/test_gpu_e2e_parity.py, tests/test_gpu_fullscale_parity.py)
SCHEDULE = {"refresh_lag": 1.0, "refresh_split": 1, "refresh_cus": -1}  # -1: fast.auto_schedule (32 CUs here since the binned sort)
# N > 1 (cadence "job": every rank's launch is 1/N of a refresh period, far shorter than the sort):
# the snapshot is sorted between launches, every rank sorting d/N of its factors
SCHEDULE_MULTI = {"refresh_lag": 0.0, "refresh_split": 1, "refresh_cus": 0}


# Parity of the schedule the default line times (STREAM, snapshot one launch older: lag 1, sort on masked CUs —
# chosen only inside the budget lr x 2 x launch <= 2,000, fast.lag_within_budget), as measured — not
# asserted — by the committed many-seed runs; the gates are in tests/test_gpu_e2e_parity.py /
# test_gpu_fullscale_parity.py / test_gpu_fullscale_reference.py.  diff = ours - reference, seed means.
PARITY_OF_TIMED_SCHEDULE = {
    "tolerance_north_star": 0.002,
    "ml20m_shape_vs_the_reference_loop_itself": {
        "source": "profiles/r05_fullepoch_reference.md (the reference's own loop imported in place, 136,677 x 20,108, d=128, "
                  "first epoch at lr 0.05: 12 / 24 / 36 / 47 refresh periods, 3 reference seeds; tests/golden/"
                  "e2e_ml20m_reference_prefix.json)",
        "ndcg@100": {"STRICT": [0.0001, -0.0001, -0.0003, 0.0008],
                     "STREAM_reference_schedule_12_seeds": [None, None, -0.0001, -0.0015],
                     "STREAM_lag1_at_lr_0.05_OUTSIDE_its_budget_12_seeds": [0.0005, 0.0014, 0.0029, -0.0065]},
        "at_this_line_lr_0.001_vs_exact_minibatches": {
            "epochs": [40, 80, 120, 160], "ndcg@100_of_STRICT": [0.1068, 0.3439, 0.4339, 0.4584],
            "STREAM_lag1": [0.0007, -0.0014, -0.0007, -0.0003], "STREAM_reference_schedule": [0.0005, 0.0011, 0.0018, 0.0013],
            "seeds": 3},
        "note": "the lagged snapshot leaves the reference's curve on the steepest part of a lr-0.05 run (three launches' "
                "worth at the end of epoch 1, +0.002 from epoch 2 on), which is why the schedule is held to the budget: "
                "this line's lr 0.001 is inside (398 <= 2,000 since r6), lr 0.01 and 0.05 are not and get the reference's schedule"},
    "small_set_vs_reference_over_epoch_orders": {
        "source": "profiles/e2e_parity_r04.txt (4,000 x 1,500 golden protocol, d=32, lr 0.05, 12 epochs; n = 200 ours, "
                  "64 reference runs over epoch orders)",
        "ndcg@100": {"epoch2": [0.0033, 21.1], "epoch4": [0.0029, 7.3], "epoch12": [0.0012, 2.5]},
        "recall@20": {"epoch2": [0.0033, 17.4], "epoch4": [0.0032, 6.3], "epoch12": [0.0012, 2.3]},
        "format": "[diff, z]",
        "note": "the older snapshot learns the steep part faster on this small set (a launch is a tenth of an "
                "epoch); the reference-schedule path reads +0.0003 / -0.0005 at epoch 12"},
    "small_set_vs_reference_fixed_order": {
        "source": "profiles/r03_e2e_many_seeds.txt (n = 200 ours, 30 reference runs)",
        "ndcg@100": {"epoch2": [0.0041, 17.3], "epoch4": [0.0032, 6.3], "epoch12": [0.0018, 3.8]},
        "recall@20": {"epoch2": [0.0038, 15.4], "epoch4": [0.0032, 4.3], "epoch12": [0.0024, 3.9]},
        "format": "[diff, z]"},
    "ml20m_shape_vs_exact_minibatches": {
        "source": "profiles/r03_fullscale_many_seeds.txt (136,677 x 20,108, d=128, lr 0.05, 6 epochs, 30 seeds "
                  "per side, against STRICT = the reference's mini-batch semantics)",
        "ndcg@100": {"epoch3": 0.0010, "epoch6": 0.0002}, "recall@20": {"epoch3": 0.0005, "epoch6": 0.0006},
        "se": 0.0004},
}


def cut_user_pieces(users, L, grouped):
    """Number of atomic user-row adds of one STREAM launch over `users` (device int32, grouped by
    user): k_stream cuts a user wherever a run boundary (a multiple of L) falls inside its triples;
    every piece of a cut user is one atomic row add, uncut users are stored plainly."""
    n = users.numel()
    if not grouped:
        return float(-(-n // L))  # every run flushes its users atomically (lower bound: one each)
    b = torch.arange(L, n, L, device=users.device)
    if b.numel() == 0:
        return 0.0
    cut = users[b - 1] == users[b]
    return float(cut.sum().item() + torch.unique(users[b[cut]]).numel())


# Hyper-parameters of the BASELINE configs (files under the reference's configs/):
# a --workload brings its own model width, batch size (refresh period), optimizer, learning rate, L2 and
# sampler; every one of them can still be overridden on the command line.
WORKLOADS = {
    # configs[2] — the config the metric is quoted on: configs/RQ2/neg-sampling/ada-sampling-ml-20m.yaml.j2
    "ml-20m": dict(dim=128, batch_size=256, optimizer="sgd", lr=0.001, reg=(0.0016, 0.0001, 0.00375),
                   sampler="adaptive", source="configs/RQ2/neg-sampling/ada-sampling-ml-20m.yaml.j2:144-153"),
    # configs[3]: configs/RQ2/neg-sampling/ada-sampling-msd.yaml.j2 (reg_alphas all: 0.00043)
    "msd": dict(dim=256, batch_size=256, optimizer="sgd", lr=0.001, reg=(0.00043, 0.00043, 0.00043),
                sampler="adaptive", source="configs/RQ2/neg-sampling/ada-sampling-msd.yaml.j2"),
    # configs[4]: configs/RQ3/time-split/ada-sampling-adam.yaml.j2:165-175
    "yelp": dict(dim=128, batch_size=256, optimizer="adam", lr=0.001, reg=(0.0025, 0.0025, 0.00025),
                 sampler="adaptive", source="configs/RQ3/time-split/ada-sampling-adam.yaml.j2:165-175"),
    # configs[1]: configs/RQ1/ours.yaml.j2:108-120 (Netflix, B = 16, uniform negatives)
    "netflix": dict(dim=64, batch_size=16, optimizer="sgd", lr=0.05, reg=(0.0025, 0.0025, 0.00025),
                    sampler="uniform", source="configs/RQ1/ours.yaml.j2:108-120"),
    # configs[0]: example.py:290-296 on the synthetic 10k x 5k set
    "cfg1-synth": dict(dim=32, batch_size=256, optimizer="sgd", lr=0.00943667980759196,
                       reg=(0.0016, 0.0001, 0.00375), sampler="uniform", source="example.py:290-296"),
}
# HBM-side bytes per k_stream launch measured by separate rocprofv3 --pmc passes (profiles/*_pmc_traffic.md)
# of exactly these commands: (workload, d, sampler, optimizer) -> file under profiles/
TRAFFIC_FILES = {
    ("ml-20m", 128, "adaptive", "sgd"): "traffic_r06.json",  # the trained state with the LDS tier (r05d: without it, early state)
    ("msd", 256, "adaptive", "sgd"): "traffic_r05_msd_d256.json",
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="ml-20m", choices=sorted(WORKLOADS))
    ap.add_argument("--dim", type=int, default=None, help="default: the workload's (WORKLOADS)")
    ap.add_argument("--sampler", choices=["adaptive", "uniform", "given"], default=None)
    ap.add_argument("--reg", type=float, nargs=3, default=None, metavar=("USER", "ITEM", "NEG"),
                    help="L2 (reg_alphas); default: the workload's")
    ap.add_argument("--optimizer", choices=["sgd", "adam", "momentum", "rmsprop"], default=None,
                    help="sgd: the fused SGD STREAM kernel (BASELINE configs[2], the default); the "
                         "others run the batched STREAM kernel (virtual mini-batches of "
                         "--batch-size, one dense torch.optim step each; BASELINE configs[4] is "
                         "--workload yelp --optimizer adam)")
    ap.add_argument("--batched", action="store_true",
                    help="run plain SGD through the batched STREAM kernel too (measurement aid)")
    ap.add_argument("--betas", type=float, nargs=2, default=(0.1, 0.999),
                    help="Adam betas (configs/RQ3/time-split/ada-sampling-adam.yaml.j2:175)")
    ap.add_argument("--adaptive-p", type=float, default=0.01)
    ap.add_argument("--lr", type=float, default=None)
    ap.add_argument("--batch-size", type=int, default=None,
                    help="reference batch size; only sets the refresh period I·ln(I)/B batches")
    ap.add_argument("--sync-every", type=int, default=1, help="item all-reduce period in steps (N>1)")
    ap.add_argument("--cadence", choices=["auto", "job", "rank"], default="auto",
                    help="N>1: how much a rank trains between two reconciliations (+ snapshot refreshes).  "
                         "'auto' (default, r4): the largest chunk that keeps lr x N x chunk inside the staleness "
                         "budget the multi-rank parity study measured (fast.STALENESS_BUDGET, "
                         "profiles/r04_cadence_study.txt) — a FULL refresh period per rank at this benchmark's "
                         "lr 0.001, period / N at lr 0.05; 'rank' = a full period per rank whatever the lr; "
                         "'job' = period / N (r3's default: the single-GPU cadence counted in job triples).  "
                         "The --tier-rows most popular rows are exchanged after every launch (two tiers)")
    ap.add_argument("--tier-rows", type=int, default=1024,
                    help="N>1, --cadence rank: rows of the hot tier (0 = one tier: r3's protocol)")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="N > 1 on ONE GPU: run the step a rank of an N-rank job runs — the chunk the cadence "
                         "picks for N ranks, the hot-tier pass after every launch, the cold fold + delta pass and "
                         "the un-fused snapshot cut per chunk — with the collectives left out (they run on a side "
                         "stream; DESIGN.md §7 models them).  Measurement aid for the parts table")
    ap.add_argument("--fuse-sync", type=int, default=1,
                    help="N>1 (or --emulate-ranks): 1 = hot-tier step + cold step + snapshot cut as one pass "
                         "(bpr_sync_cut); 0 = four kernels")
    ap.add_argument("--hot-split", type=int, default=1,
                    help="N>1, hot tier: launches per step, with a hot-tier exchange after each")
    ap.add_argument("--no-shard-refresh", action="store_true",
                    help="N>1, --refresh-lag 0: every rank sorts ALL factors of the snapshot instead of d/N of "
                         "them + an all-gather (Engine.adaptive_refresh_sharded, the default)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink users/actions (debug)")
    ap.add_argument("--max-inflight", type=int, default=0)
    ap.add_argument("--time-every", type=int, default=8,
                    help="hipEvent timing of the dominant kernel on every N-th launch of the timed region")
    ap.add_argument("--run-len", type=int, default=0, help="0: chosen by the library from the launch size")
    ap.add_argument("--refresh-lag", type=float, default=None,
                    help="adaptive snapshot schedule (DESIGN.md §4.3): 0 = sorted between launches "
                         "(the reference's update_stats), 1 = cut before the previous launch and "
                         "sorted beside it on the side stream, 0<f<1 = cut at 1-f of the previous "
                         "launch.  Default: the schedule the parity gates hold (see SCHEDULE)")
    ap.add_argument("--refresh-split", type=int, default=None,
                    help="launches (= steps) per refresh period, snapshot retaken for each")
    ap.add_argument("--refresh-cus", type=int, default=None,
                    help="CUs (of 256) the side stream's sort is masked to; the STREAM kernel runs "
                         "on the complementary mask (0: unmasked streams; -1: by shape, "
                         "fast.auto_refresh_cus — 64 for the ML-20M workload)")
    ap.add_argument("--main-cus", type=int, default=0,
                    help="measurement aid: run everything on a stream masked to the LAST N CUs")
    ap.add_argument("--hot-rows", type=int, default=None,
                    help="delta rows for the N most popular item rows (library default 256; 0 = off)")
    ap.add_argument("--hot-replicas", type=int, default=1)
    ap.add_argument("--item-skew", type=float, default=None,
                    help="override the item popularity exponent (debug: 0 = uniform popularity)")
    ap.add_argument("--ungrouped", action="store_true", help="all-atomic user rows (debug)")
    ap.add_argument("--jit-plan", type=int, default=0,
                    help="1: with the overlapped snapshot schedule the epoch is never planned as a whole — every "
                         "chunk is planned by bpr_plan_chunk on the side stream, one step ahead (measured SLOWER: "
                         "624 M vs 748 M triples/s, profiles/r04_jit_plan.md); 0 (default): bpr_plan_epoch at "
                         "every epoch boundary")
    ap.add_argument("--plan-ahead", type=int, default=0,
                    help="1: with the snapshot sort on masked CUs, the NEXT epoch's bpr_plan_epoch runs on a third stream "
                         "masked to the sorter's CUs while this epoch trains (same plan, other buffers).  Measured and "
                         "lost (profiles/r05b_plan_ahead.txt: 711 against 799 M triples/s — the plan's radix sort holds the "
                         "sorter's CUs for milliseconds and the snapshot sort becomes the critical path); 0 (default): "
                         "between the epochs on the launch stream")
    ap.add_argument("--async-cut", type=int, default=-1,
                    help="1: the transpose of the next snapshot's keys runs on the side stream beside the next launch "
                         "(bpr_train_stream_acut; the fold of the hot block stays on the launch stream) instead of "
                         "between two launches; -1 (default): as fast.StreamTrainer's auto — on with the overlapped "
                         "schedule on masked streams, one GPU; 0: off")
    ap.add_argument("--item-bias", type=int, default=0,
                    help="1: the model carries the reference's optional item_bias (models/bpr/model.py:101-110; "
                         "its RQ configs switch it on).  Single GPU only here (a measurement aid: "
                         "profiles/shapes_r04.txt); the default 0 is the model every earlier round timed")
    ap.add_argument("--sustained-epochs", type=int, default=3,
                    help="after the K-step region: this many WHOLE epochs (every plan and every step in place) timed "
                         "by wall clock between barriers — the headline `value`, on tables a few epochs from their "
                         "random init as every round's line was (-1: as many as make ~0.6 s of work; 0 = skip: "
                         "`value` is then the K-step region's).  See --steady-epochs for the trained state")
    ap.add_argument("--steady-epochs", type=int, default=30,
                    help="after the sustained region the job trains on, untimed, until this many epochs are done, and "
                         "ONE more epoch is timed by wall clock: `steady_state` (0 = skip).  The adaptive sampler's "
                         "negatives concentrate on popular rows and its walks get deeper once the model has moved "
                         "(DESIGN.md §4.1 r5; tools/trained_state_probe.py): a long job sees this number, the first "
                         "epochs from random init see `value`")
    ap.add_argument("--steady-timed-epochs", type=int, default=100,
                    help="whole epochs timed for the steady state — the headline `value` (r6): epochs steady-epochs + 1 .. "
                         "steady-epochs + this many of the same job, by wall clock between barriers")
    ap.add_argument("--launch-split", type=int, default=0,
                    help="launches per refresh period that read the same snapshot (fast.StreamTrainer launch_split); 0 = "
                         "auto: 2 outside the one-rank budget lr x 2 x period <= 2,000, else 1")
    ap.add_argument("--hot-lds", type=int, default=-1,
                    help="rows of the hot block a CU keeps in LDS during a launch (bpr_set_hot_lds, r6): -1 = by the "
                         "staleness budget (fast.hot_lds_rows: on at lr 0.001 / 0.01, off at 0.05), 0 = off, n = forced")
    ap.add_argument("--partial-snapshot", type=int, default=0,
                    help="1: the split refresh sorts only the two ends of every snapshot column and buckets the middle "
                         "(bpr_set_tuning partial_snapshot; DESIGN.md §4.3 r5)")
    ap.add_argument("--partial-target", type=int, default=640)
    ap.add_argument("--force-dist", action="store_true",
                    help="WORLD_SIZE = 1 under torch.distributed.run: run the N > 1 code path anyway — RCCL process "
                         "group, two-tier ItemSync, fused sync-cut, sharded refresh — with one rank (de-risks the "
                         "first node run; tests/test_gpu_bench.py)")
    ap.add_argument("--split-cold-mb", type=float, default=0.0,
                    help="N>1: all-reduces of at least this many MB run as reduce-scatter + all-gather (0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--seed", type=int, default=13)
    args = ap.parse_args()
    w = WORKLOADS[args.workload]
    for k in ("dim", "sampler", "optimizer", "lr", "batch_size"):
        if getattr(args, k) is None:
            setattr(args, k, w[k])
    args.reg = tuple(args.reg) if args.reg is not None else tuple(w["reg"])
    return args


def cpu_baseline(data, d, reg, lr, p, seconds, seed, sampler="adaptive", B=256):
    """SURVEY §8d baseline (a): the C restatement of the same path (the oracle: sample a mini-batch's
    negatives -> B gradients at the pre-step parameters -> one sparse SGD step; adaptive snapshot retaken
    every I ln I / B batches) with OpenMP over the triples of a batch, timed on this host's cores over a
    bounded number of batches.  The thread count is the fastest of a short probe; both are stated."""
    import oracle

    host = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    rng = np.random.default_rng(seed)
    P = ((rng.random((data.num_users, d)) - 0.5) / d).astype(np.float32)
    Q = ((rng.random((data.num_items, d)) - 0.5) / d).astype(np.float32)
    P[0] = 0
    Q[0] = 0
    QT, sigma = oracle.adaptive_stats(Q)
    order = oracle.adaptive_order(QT)
    perm = rng.permutation(data.nnz)
    users = np.ascontiguousarray(data.users[perm])
    items = np.ascontiguousarray(data.items[perm])
    every = max(1, int(data.num_items * math.log(data.num_items) / B))
    smp = oracle.NEG_ADAPTIVE if sampler == "adaptive" else oracle.NEG_UNIFORM

    def run(lo, secs, threads):
        t0 = time.perf_counter()
        done, _ = oracle.train_batches_omp(P, Q, users[lo:], items[lo:], B, smp, lr, reg, adaptive_p=p, QT=QT,
                                           sigma=sigma, order=order, refresh_every=every, indptr=data.indptr,
                                           indices=data.indices, seed=seed, offset=lo, seconds=secs, threads=threads)
        return done, time.perf_counter() - t0

    probe, lo = {}, 0
    for th in sorted({1, min(host, 8), min(host, 32), min(host, 64), host}):
        done, dt = run(lo, 0.5, th)
        probe[th] = done / dt
        lo += done
    threads = max(probe, key=probe.get)
    done, dt = run(lo, seconds, threads)
    return {
        "value": done / dt, "unit": "triples/s", "cores": threads, "kind": "port",
        "sample": f"{done // B} batches of {B} triples ({done} triples, {dt:.1f} s) of the same workload: the C "
                  f"restatement (oracle/bpr_oracle.c orc_train_batches_omp: {sampler} sampler + strict SGD step, "
                  f"snapshot refresh every {every} batches included) with OpenMP over the triples of a batch, "
                  f"{threads} threads = the fastest of {sorted(probe)} on this {host}-core host "
                  f"({', '.join(f'{k}: {v / 1e3:.0f} k/s' for k, v in sorted(probe.items()))})",
    }


def torch_optimizer(params, name, lr, betas):
    """the torch.optim object the reference's config would build for --optimizer (bench defaults:
    momentum 0.9, RMSprop alpha 0.99 — what Engine.set_optimizer is given in main())"""
    if name == "adam":
        return torch.optim.Adam(params, lr=lr, betas=tuple(betas))
    if name == "momentum":
        return torch.optim.SGD(params, lr=lr, momentum=0.9)
    if name == "rmsprop":
        return torch.optim.RMSprop(params, lr=lr, alpha=0.99)
    return torch.optim.SGD(params, lr=lr)


def cpu_reference_op_sequence(data, d, reg, lr, p, sampler, seed, warm=20, reps=3, steps=8, optimizer="sgd",
                              betas=(0.9, 0.999), B=256):
    """(b) of SURVEY §8d: the REFERENCE's op sequence restated on CPU tensors and timed with every
    host core — per batch of 256: `_sampling_weights` ([B, I] weights, seen + item 0 zeroed, row
    normalised: revisit_bpr/modules/neg_samplers.py:135-141) -> `torch.multinomial` (:31-37) or the
    adaptive path ([B, I] gather of the snapshot, scatter -1e13, full argsort: :74-124) -> dense
    autograd forward / backward (models/bpr/model.py:48-68, set_backend("torch") here) -> dense
    `torch.optim.SGD.step` over both tables (example.py:176-180).  >= 20 warm steps, median of 3
    timed blocks."""
    from revisit_bpr.models import BPR
    from revisit_bpr.models.bpr import MF, set_backend

    host = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    U, I = data.num_users, data.num_items
    torch.manual_seed(seed)
    model = BPR(fuse_forward=True, reg_alphas=dict(zip(("user", "item", "neg"), reg)),
                logits_model=MF(torch.nn.Embedding(U, d, padding_idx=0),
                                torch.nn.Embedding(I, d, padding_idx=0)))
    opt = torch_optimizer(model.parameters(), optimizer, lr, betas)
    gen = torch.Generator().manual_seed(seed)
    lens = np.diff(data.indptr)
    S = int(min(lens.max(), 4096))
    rng = np.random.default_rng(seed)
    perm = rng.permutation(data.nnz)
    users_t = torch.from_numpy(data.users.astype(np.int64))
    items_t = torch.from_numpy(data.items.astype(np.int64))
    ones = torch.ones(I)
    feats = model.logits_model.get_features()
    snap_T = feats["item"].detach().t().clone()                  # update_stats (:126-132)
    snap_std = feats["item"][1:].detach().std(dim=0, keepdim=True)

    def seen_of(u):  # the padded [B, S] seen matrix the reference's collator hands over
        out = np.zeros((len(u), S), np.int64)
        for r, uu in enumerate(u):
            row = data.indices[data.indptr[uu]:data.indptr[uu + 1]][:S]
            out[r, :len(row)] = row
        return torch.from_numpy(out)

    def one_step(k):
        idx = perm[k * B:(k + 1) * B]
        u = users_t[idx]
        seen = seen_of(data.users[idx])
        w = ones.expand(len(idx), -1).scatter(-1, seen, 0.0)      # _sampling_weights
        w[:, 0] = 0.0
        w = w * w.sum(-1, keepdim=True).reciprocal()
        if sampler == "adaptive":
            n_unseen = w.gt(0).sum(-1, keepdim=True)
            pu = feats["user"].detach()[u]
            factor = torch.multinomial(pu.abs() * snap_std, 1, generator=gen)
            rank = torch.empty_like(factor).geometric_(p, generator=gen).clamp_(max=n_unseen)
            rank = torch.where(pu.gather(-1, factor).gt(0), rank - 1, n_unseen - rank)
            seen0 = torch.hstack((seen, torch.zeros_like(rank)))
            neg = torch.argsort(-snap_T[factor.squeeze(-1)].scatter(-1, seen0, -1e13), dim=-1) \
                .gather(-1, rank)
        else:
            neg = torch.multinomial(w, 1, generator=gen)
        out = model({"user": u, "item": items_t[idx].unsqueeze(-1), "neg": neg})
        out["loss"].backward()
        opt.step()
        opt.zero_grad()

    set_backend("torch")
    try:
        model.train()
        # every host core is offered; torch's intra-op pool is slower with hundreds of threads on
        # these op sizes than with a few dozen, so the thread count is the fastest of a short probe
        probe = {}
        for th in sorted({min(host, 8), min(host, 32), min(host, 64), host}):
            torch.set_num_threads(th)
            one_step(0)
            t0 = time.perf_counter()
            for k in range(2):
                one_step(1 + k)
            probe[th] = (time.perf_counter() - t0) / 2
        threads = min(probe, key=probe.get)
        torch.set_num_threads(threads)
        for k in range(warm):
            one_step(k)
        times = []
        for r in range(reps):
            t0 = time.perf_counter()
            for k in range(steps):
                one_step(warm + r * steps + k)
            times.append((time.perf_counter() - t0) / steps)
    finally:
        set_backend("hip")
    ms = float(np.median(times)) * 1e3
    return {
        "value": B / (ms * 1e-3), "unit": "triples/s", "cores": threads,
        "kind": "reference-op-sequence",
        "sample": f"the reference's op sequence restated on CPU tensors (torch {torch.__version__}, "
                  f"{threads} threads = the fastest of {sorted(probe)} on this {host}-core host): "
                  f"[{B}, I] sampling weights + "
                  f"{'adaptive argsort' if sampler == 'adaptive' else 'multinomial'}, dense autograd, "
                  f"dense {type(opt).__module__.split('.')[-1]}.{type(opt).__name__} step; {warm} warm steps, "
                  f"median of {reps} x {steps} timed steps "
                  f"({ms:.1f} ms per step of {B} triples) of the same workload",
    }


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libbprcore has no CPU path")
    # one rank per GPU; the modulo only matters for functional tests of the N>1 path on a 1-GPU box
    # (BPR_DIST_BACKEND=gloo, several ranks sharing cuda:0) — RCCL itself needs one device per rank
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    forced = bool(args.force_dist) and world == 1  # the N > 1 code path with ONE rank, through RCCL
    if forced and "MASTER_ADDR" not in os.environ:
        raise SystemExit("--force-dist needs the rendezvous of torch.distributed.run (--nproc-per-node 1)")
    rccl_ranks_seen = None
    if world > 1 or forced:
        backend = os.environ.get("BPR_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        # ranks the communicator really spans: every rank contributes a one
        ones = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        rccl_ranks_seen = int(ones.item())
        assert rccl_ranks_seen == world, (rccl_ranks_seen, world)

    from revisit_bpr import engine as eng
    from revisit_bpr.datasets import synthetic
    from revisit_bpr.distributed import ItemSync

    d = args.dim
    # every rank owns its own ML-20M-shaped user shard (weak scaling); same item space
    gen_kw = {} if args.item_skew is None else {"item_skew": args.item_skew}  # measurement aid
    data = synthetic.generate_named(args.workload, eval_users=10_000, seed=args.seed + rank,
                                    scale=args.scale, **gen_kw)
    U, I = data.num_users, data.num_items
    g = torch.Generator().manual_seed(args.seed)  # same Q on every rank
    Q = ((torch.rand(I, d, generator=g) - 0.5) / d)
    gp = torch.Generator().manual_seed(args.seed + 1000 + rank)
    P = ((torch.rand(U, d, generator=gp) - 0.5) / d)
    P[0] = 0
    Q[0] = 0
    P, Q = P.to(dev), Q.to(dev)
    if args.item_bias and (world > 1 or args.emulate_ranks > 1):
        raise SystemExit("--item-bias 1 is a single-GPU measurement")
    item_bias = torch.zeros(I, device=dev) if args.item_bias else None
    e = eng.Engine(P, Q, item_bias)
    reg = args.reg  # the workload's reg_alphas (WORKLOADS) unless --reg
    e.set_reg(*reg)
    batched = args.batched or args.optimizer != "sgd"
    kind = {"sgd": eng.OPT_SGD, "momentum": eng.OPT_MOMENTUM, "adam": eng.OPT_ADAM,
            "rmsprop": eng.OPT_RMSPROP}[args.optimizer]
    e.set_optimizer(kind, lr=args.lr, momentum=0.9, betas=tuple(args.betas), alpha=0.99)
    opt_state = e.alloc_opt_state()  # noqa: F841  (keeps the state tensors alive)
    e.bind_seen_csr(torch.from_numpy(data.indptr).to(dev), torch.from_numpy(data.indices).to(dev))
    sampler = {"adaptive": eng.NEG_ADAPTIVE, "uniform": eng.NEG_UNIFORM, "given": eng.NEG_GIVEN}[args.sampler]

    # snapshot schedule of the adaptive sampler
    sched = SCHEDULE if (world == 1 and not forced) or args.cadence != "job" else SCHEDULE_MULTI
    lag = sched["refresh_lag"] if args.refresh_lag is None else args.refresh_lag
    split = sched["refresh_split"] if args.refresh_split is None else args.refresh_split
    cus = sched["refresh_cus"] if args.refresh_cus is None else args.refresh_cus
    shard_refresh = (world > 1 or forced) and lag == 0.0 and not args.no_shard_refresh
    if sampler != eng.NEG_ADAPTIVE or batched:
        lag, split, cus = 0.0, 1, 0
    # epoch order: bpr_plan_epoch = seeded pseudo-random partition of the triple list into chunks of
    # one launch, each grouped by user (the STREAM kernel keeps the user row in registers)
    every = max(1, int(I * math.log(I) / args.batch_size))  # example.py:302
    period = min(every * args.batch_size, data.nnz)
    from revisit_bpr.fast import launches_per_period
    emu = args.emulate_ranks if (args.emulate_ranks > 1 and world == 1) else 0
    cad_world = emu if emu else world
    ranks_per_period = (cad_world if args.cadence == "job" else 1 if args.cadence == "rank" else
                        launches_per_period(args.lr, cad_world, period))
    # r6, as fast.StreamTrainer(launch_split="auto"): outside the one-rank budget (high learning rates) a period runs as
    # TWO launches that read the same snapshot, so that a user's triples of a period are not applied back to back
    from revisit_bpr.fast import lag_within_budget
    lsplit = args.launch_split if args.launch_split > 0 else (
        1 if (cad_world > 1 or split != 1 or batched or (args.refresh_lag or 0.0) != 0.0
              or lag_within_budget(args.lr, period)) else 2)
    chunk = max(1, period // (split * ranks_per_period * lsplit))
    n_chunks = max(1, data.nnz // chunk)
    if world > 1:  # every rank's own shard: the same number of steps per epoch everywhere, or the collectives of
        tn = torch.tensor([n_chunks], device=dev)  # a sustained / steady epoch stop matching up
        dist.all_reduce(tn, op=dist.ReduceOp.MIN)
        n_chunks = int(tn.item())
    src_users = torch.from_numpy(data.users).to(dev)
    src_items = torch.from_numpy(data.items).to(dev)
    users_e, items_e = torch.empty_like(src_users), torch.empty_like(src_items)
    users_sorted = eng.Engine.users_sorted(src_users)  # the triple list in CSR order: bpr_plan_epoch takes one radix pass
    e.set_stream_opts(not args.ungrouped, args.run_len)
    e.set_bias_tracking(True)  # this loop owns the item_bias between its launches (as fast.StreamTrainer's does)
    if args.partial_snapshot:
        e.set_tuning("partial_snapshot", 1)
        e.set_tuning("partial_target", args.partial_target)
    if args.hot_rows is not None:
        e.set_hot_rows(args.hot_rows, args.hot_replicas)
    from revisit_bpr.fast import hot_lds_rows
    lds_rows_asked = hot_lds_rows(args.lr, chunk, cad_world) if args.hot_lds < 0 else args.hot_lds
    if not batched:
        e.set_hot_lds(lds_rows_asked)
    main_stream = side_stream = None
    if lag > 0.0 and cus != 0:
        total_cus = torch.cuda.get_device_properties(dev).multi_processor_count
        if cus < 0:
            # by shape (fast.auto_schedule: 32 CUs for the ML-20M workload, 96 for MSD d=256); it may
            # also say that no split of the chip beats the reference's serial schedule
            from revisit_bpr.fast import auto_schedule
            a_lag, cus = auto_schedule(I, d, chunk, total_cus, lr=args.lr)  # (lr: the staleness budget of a lagged snapshot)
            if a_lag == 0.0 and args.refresh_lag is None:
                lag, cus = 0.0, 0
        if cus > 0:
            side_stream = eng.MaskedStream(dev, eng.cu_mask(0, cus, total_cus))
            main_stream = eng.MaskedStream(dev, eng.cu_mask(cus, total_cus - cus, total_cus))
            e.set_side_stream(side_stream)
    if main_stream is None and args.main_cus > 0:
        total_cus = torch.cuda.get_device_properties(dev).multi_processor_count
        main_stream = eng.MaskedStream(dev, eng.cu_mask(total_cus - args.main_cus, args.main_cus, total_cus))
    sync = None
    if world > 1 or emu or forced:
        tier = args.tier_rows if (args.cadence != "job" and not batched) else 0
        comm = None
        if forced or args.split_cold_mb > 0:
            from revisit_bpr.distributed import _DistComm
            comm = _DistComm(None, True, force=forced, split_bytes=int(args.split_cold_mb * 2**20))
        sync = ItemSync([Q], engine=e, hot_rows=tier, local_items=src_items, force_tiers=bool(emu) or forced,
                        comm=comm)
    pieces = args.hot_split if (sync is not None and sync.hot_tier) else 1
    scalars = torch.zeros(4, device=dev)
    seed = args.seed
    given_neg = (torch.randint(1, I, (chunk,), device=dev, dtype=torch.int32)
                 if sampler == eng.NEG_GIVEN else None)  # measurement aid only

    fused = lag >= 1.0 and world == 1 and not emu and not forced  # the launch's epilogue cuts the next snapshot's keys
    # several ranks: the reconciliation passes and the cut are ONE pass after the launch (--fuse-sync 0: r4's
    # first version, four kernels)
    fused_sync = (lag >= 1.0 and sync is not None and sync.hot_tier and sync.can_fuse and args.sync_every == 1
                  and bool(args.fuse_sync))
    synced = [False]
    # --async-cut: the snapshot cut leaves the launch stream (bpr_train_stream_acut: a read-only pass on the
    # side stream beside the next launch); the hot rows are folded before the tables are read
    from revisit_bpr.fast import auto_async_cut
    acut = (bool(args.async_cut) if args.async_cut >= 0 else (side_stream is not None and auto_async_cut(I, cus))) and fused

    # --jit-plan (default with the overlapped schedule): no bpr_plan_epoch at all — chunk k + 1 is
    # planned by bpr_plan_chunk on the side stream behind the sort of step k (the plan does not depend
    # on the model), into the buffer launch k - 1 read; bpr_adaptive_refresh_commit waits for both
    jit = bool(args.jit_plan) and sampler == eng.NEG_ADAPTIVE and lag >= 1.0 and not batched
    cbuf = [(torch.empty(chunk, dtype=torch.int32, device=dev), torch.empty(chunk, dtype=torch.int32, device=dev))
            for _ in range(2)] if jit else None
    planned = set()
    # --plan-ahead: epoch e + 1 is planned beside epoch e, on the sorter's CUs (fast.StreamTrainer does the same)
    plan_ahead = bool(args.plan_ahead) and side_stream is not None and not jit and not batched
    ebuf = [(users_e, items_e)]
    cur, pre, plan_stream = [0], {}, None
    if plan_ahead:
        ebuf.append((torch.empty_like(src_users), torch.empty_like(src_items)))
        # --plan-ahead 1: on the sorter's CUs (r5: lost; r6 with the one-pass plan: 777 against 830 M); 2 (r6): on the LAUNCH
        # stream's CUs — the 1,024-thread LDS-tier workgroups leave half of a CU's wave slots free and the plan's kernels
        # use no LDS: measured NEUTRAL, 827-830 against 830 M (the launches beside the plan slow down by what the plan
        # saves: profiles/r06_hotlds.md)
        plan_stream = eng.MaskedStream(dev, eng.cu_mask(0, cus, total_cus) if args.plan_ahead == 1 else
                                       eng.cu_mask(cus, total_cus - cus, total_cus))

    def plan_chunk(kk: int, on_side: bool):
        e.plan_chunk(src_users, src_items, chunk, seed + kk // n_chunks, kk % n_chunks, out=cbuf[kk & 1],
                     on_side=on_side)

    def launch(k: int, lo: int, hi: int, base: int, cut: bool = False):
        users, items = (cbuf[k & 1][0], cbuf[k & 1][1]) if jit else ebuf[cur[0]]
        if jit:  # a chunk buffer: positions relative to the chunk
            lo, hi, base = lo - base, hi - base, 0
        if batched:
            e.train_stream_batched(users[lo:hi], items[lo:hi], args.batch_size,
                                   sampler=sampler, neg=given_neg, adaptive_p=args.adaptive_p,
                                   seed=seed, offset=(rank << 40) + k * chunk + (lo - base),
                                   max_inflight=args.max_inflight, scalars=scalars)
        else:
            for p in range(pieces):  # hot tier: an exchange of the hot block after every launch
                a, b = lo + (hi - lo) * p // pieces, lo + (hi - lo) * (p + 1) // pieces
                e.train_stream(users[a:b], items[a:b], sampler=sampler,
                               neg=None if given_neg is None else given_neg[:b - a],
                               adaptive_p=args.adaptive_p, seed=seed,
                               offset=(rank << 40) + k * chunk + (a - base),
                               max_inflight=args.max_inflight, scalars=scalars,
                               cut=(("async" if acut else True) if (cut and p == pieces - 1) else False))
                if fused_sync and cut and p == pieces - 1:
                    sync.step_cut()  # hot step + cold step + snapshot cut: one pass (bpr_sync_cut)
                    synced[0] = True
                elif sync is not None and sync.hot_tier:
                    sync.hot_step()

    def step(k: int):
        c = k % n_chunks
        lo = c * chunk
        if jit:
            if k not in planned:  # the first step, or a jump: plan it on the launch stream
                plan_chunk(k, False)
        elif c == 0:  # new epoch: re-plan (part of the job)
            if batched:
                e.shuffle_epoch(src_users, src_items, seed + k // n_chunks, out=(users_e, items_e))
            else:
                ep = k // n_chunks
                if ep in pre:  # planned beside the previous epoch
                    cur[0], ev = pre.pop(ep)
                    torch.cuda.current_stream().wait_event(ev)
                else:
                    e.plan_epoch(src_users, src_items, chunk, seed + ep, out=ebuf[cur[0]], sorted_input=users_sorted)
                if plan_ahead:
                    nxt = cur[0] ^ 1
                    ev0 = torch.cuda.Event()
                    ev0.record()  # the launches that read that buffer last (the previous epoch's) are all queued
                    with torch.cuda.stream(plan_stream.torch):
                        plan_stream.torch.wait_event(ev0)
                        e.plan_epoch(src_users, src_items, chunk, seed + ep + 1, out=ebuf[nxt], sorted_input=users_sorted)
                        ev = torch.cuda.Event()
                        ev.record()
                    e._sync_stream()  # the library back on this stream
                    pre.clear()
                    pre[ep + 1] = (nxt, ev)
        if sampler != eng.NEG_ADAPTIVE:
            launch(k, lo, lo + chunk, lo)
        elif lag == 0.0:
            if shard_refresh and not batched:
                e.adaptive_refresh_sharded(rank, world, force=forced)
            elif c % lsplit == 0:  # (launch_split: the period's later launches reuse the snapshot)
                e.adaptive_refresh()  # batched: brings the item rows to "now" first
            launch(k, lo, lo + chunk, lo)
        else:  # the same schedule as fast.StreamTrainer._chunk
            if e.refresh_pending():
                e.adaptive_refresh_commit()
            else:
                e.adaptive_refresh()
            cut = lo if lag >= 1.0 else lo + max(1, int(round((1.0 - lag) * chunk)))
            if cut > lo:
                launch(k, lo, cut, lo)
            e.adaptive_refresh_begin()
            if jit:
                plan_chunk(k + 1, True)
                planned.clear()
                planned.add(k + 1)
            if cut < lo + chunk:
                launch(k, cut, lo + chunk, lo, cut=fused or fused_sync)
        if sync is not None and (k + 1) % args.sync_every == 0 and not synced[0]:
            if batched:
                e.flush_items()
            sync.step()
        synced[0] = False

    def barrier():
        torch.cuda.synchronize()
        if world > 1 or forced:
            dist.barrier()
        torch.cuda.synchronize()

    import contextlib
    on_main = torch.cuda.stream(main_stream.torch) if main_stream is not None else contextlib.nullcontext()
    q_before = Q.double().sum().item(), Q.abs().double().sum().item()
    torch.cuda.synchronize()
    # The timed region is K consecutive steps in their natural order (step k = chunk k mod n_chunks;
    # an epoch = n_chunks steps and starts with a bpr_plan_epoch).  With the driver's short regions
    # no epoch boundary may fall inside: the plan is then timed separately (after the region) and
    # its amortised share — K / n_chunks plans — is ADDED to the measured time, never subtracted.
    k0 = 0
    with on_main:
        for k in range(k0, k0 + args.warmup + 1):
            step(k)
        barrier()
        # hipEvents around every `--time-every`-th launch of the dominant kernel (each timed launch
        # idles the stream for ~12 us; the rocprofv3 summary under profiles/ times all of them)
        e.timing_enable(max(1, args.time_every))
        if sync is not None:
            sync.timing = True
        scalars.zero_()
        t0 = time.perf_counter()
        first = k0 + args.warmup + 1
        for k in range(first, first + args.steps):
            step(k)
        if sync is not None:
            sync.hot_finish()
            sync.finish()
        if acut:
            e.hot_fold()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        barrier()
        # SUSTAINED: whole epochs in place — every epoch's bpr_plan_epoch where it belongs, nothing
        # modelled — wall clock over `sus_epochs` x n_chunks steps (the driver's 20-step
        # region is 5 ms and holds no plan)
        sus_epochs = args.sustained_epochs
        if sus_epochs < 0:  # ~0.6 s of work (the K-step region of the driver's --steps 20 is a few ms), >= 3 epochs
            sus_epochs = int(min(max(math.ceil(0.6 / max(dt / args.steps * n_chunks, 1e-9)), 3), 400))
            if world > 1:  # the same count on every rank
                te = torch.tensor([sus_epochs], device=dev)
                dist.all_reduce(te, op=dist.ReduceOp.MAX)
                sus_epochs = int(te.item())
        sus_dt = 0.0
        if sus_epochs > 0:
            k_s = ((first + args.steps) // n_chunks + 1) * n_chunks  # the next epoch boundary
            ts = time.perf_counter()
            for k in range(k_s, k_s + sus_epochs * n_chunks):
                step(k)
            if sync is not None:
                sync.hot_finish()
                sync.finish()
            if acut:
                e.hot_fold()
            torch.cuda.synchronize()
            sus_dt = time.perf_counter() - ts
            barrier()
        kernel_ms, launches = e.timing_read()
        e.timing_enable(False)
        # STEADY STATE: train on (untimed) to --steady-epochs epochs, then one whole epoch by wall clock
        steady = None
        steady_steps = 0
        if args.steady_epochs > 0:
            k_now = (k_s + sus_epochs * n_chunks) if sus_epochs > 0 else ((first + args.steps) // n_chunks + 1) * n_chunks
            k_end = max(args.steady_epochs * n_chunks, k_now)
            for k in range(k_now, k_end):
                step(k)
            barrier()
            e.timing_enable(max(1, args.time_every))
            st_epochs = max(1, args.steady_timed_epochs)
            tq = time.perf_counter()
            for k in range(k_end, k_end + n_chunks):  # the first timed epoch on its own: r5's `steady_state` was this epoch
                step(k)
            torch.cuda.synchronize()
            st_first = time.perf_counter() - tq
            for k in range(k_end + n_chunks, k_end + st_epochs * n_chunks):
                step(k)
            if sync is not None:
                sync.hot_finish()
                sync.finish()
            if acut:
                e.hot_fold()
            torch.cuda.synchronize()
            st_dt = time.perf_counter() - tq
            barrier()
            st_kernel_ms, st_launches = e.timing_read()
            e.timing_enable(False)
            steady_steps = k_end + st_epochs * n_chunks - k_now
            if world > 1:
                tt = torch.tensor([st_dt], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                st_dt = float(tt.item())
            steady = {"epochs_trained_before": k_end // n_chunks, "epochs": st_epochs, "steps": st_epochs * n_chunks,
                      "ms_per_step": st_dt * 1e3 / (st_epochs * n_chunks),
                      "value": st_epochs * n_chunks * chunk * world / st_dt, "kernel_ms_avg": st_kernel_ms,
                      "kernel_launches_timed": st_launches,
                      "first_epoch": {"epoch": k_end // n_chunks + 1, "ms_per_step": st_first * 1e3 / n_chunks,
                                      "value": n_chunks * chunk * world / st_first,
                                      "note": "the first of the timed epochs alone (what r5 reported as steady_state); rank 0's clock"},
                      "note": "whole epochs by wall clock (every bpr_plan_epoch in place, nothing modelled) after that many "
                              "epochs of the same job (lr as configured): the state a long training run is in"}
        # the epoch plan, timed on its own (3 calls; it does not touch the model)
        tp = time.perf_counter()
        for r in range(3):
            if batched:
                e.shuffle_epoch(src_users, src_items, seed + 1000 + r, out=(users_e, items_e))
            else:
                e.plan_epoch(src_users, src_items, chunk, seed + 1000 + r, out=(users_e, items_e), sorted_input=users_sorted)
        torch.cuda.synchronize()
        plan_ms = (time.perf_counter() - tp) * 1e3 / 3
    plans_timed = sum(1 for k in range(first, first + args.steps) if k % n_chunks == 0)
    dt_measured = dt
    if jit:  # every chunk's plan ran inside the region (on the side stream): nothing to add
        plans_timed = args.steps / n_chunks
    dt += max(0.0, args.steps / n_chunks - plans_timed) * plan_ms * 1e-3
    if world > 1:
        t = torch.tensor([dt, sus_dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, sus_dt = float(t[0].item()), float(t[1].item())
    sc = scalars.cpu().numpy()
    if batched:
        e.flush_lazy()
    q_after = Q.double().sum().item(), Q.abs().double().sum().item()
    # a kernel that did nothing cannot produce a number: every triple of the timed region was
    # counted by the kernel itself, the loss it accumulated is a real -log sigma, the tables moved
    counted = (args.steps + sus_epochs * n_chunks + steady_steps) * chunk
    assert int(round(float(sc[3]))) == counted, (sc[3], counted)
    assert 0.0 < float(sc[0] / sc[3]) < 5.0 and q_after != q_before and torch.isfinite(Q).all()
    # line-atomics per triple of the chunk planned last (what the L2 atomic units see): 2 item rows
    # of d*4/128 lines each + one row per piece of every user a run boundary cuts (k_stream's rule)
    lines_per_row = max(1, (d * 4) // 128)
    user_atomic_rows = 0.0
    if not batched:
        L = e.stream_run_len()  # what the library's last launch used (it picks it from the occupancy)
        planned_users = cbuf[(first + args.steps - 1) & 1][0] if jit else users_e[:chunk]
        user_atomic_rows = cut_user_pieces(planned_users, L, not args.ungrouped) / chunk

    opt_desc = {"sgd": f"SGD lr={args.lr}", "momentum": f"SGD(momentum 0.9) lr={args.lr}",
                "adam": f"Adam lr={args.lr} betas={tuple(args.betas)}",
                "rmsprop": f"RMSprop lr={args.lr} alpha=0.99"}[args.optimizer]
    if rank == 0:
        triples = args.steps * chunk * world
        region_value = args.steps * chunk * world / dt_measured  # the K-step region as measured (no plan inside unless an epoch began there)
        region_value_with_plan = triples / dt
        # headline (r6): the TRAINED state — whole epochs by wall clock after `--steady-epochs` epochs
        # of the same job (every plan in place, nothing modelled) — when that leg ran; else the early whole epochs
        # (`sustained`); else the K-step region.  The early number rides along as `early_state`.
        early_value = (sus_epochs * n_chunks * chunk * world / sus_dt) if sus_epochs > 0 else region_value_with_plan
        early_ms = (sus_dt * 1e3 / (sus_epochs * n_chunks)) if sus_epochs > 0 else dt * 1e3 / args.steps
        early_kernel_ms = kernel_ms
        if steady is not None:
            value, ms_per_step, steps_behind = steady["value"], steady["ms_per_step"], steady["steps"]
            kernel_ms, launches = steady["kernel_ms_avg"], steady["kernel_launches_timed"]
            value_source = ("steady state: epochs %d..%d of the job (%d whole epochs = %d steps) by wall clock between "
                            "barriers, every bpr_plan_epoch in place, nothing modelled" %
                            (steady["epochs_trained_before"] + 1, steady["epochs_trained_before"] + steady["epochs"],
                             steady["epochs"], steady["steps"]))
        elif sus_epochs > 0:
            value, ms_per_step, steps_behind = early_value, early_ms, sus_epochs * n_chunks
            value_source = ("sustained: %d whole epochs = %d steps from random init by wall clock between barriers, every "
                            "bpr_plan_epoch in place, nothing modelled" % (sus_epochs, sus_epochs * n_chunks))
        else:
            value, ms_per_step, steps_behind = region_value_with_plan, dt * 1e3 / args.steps, args.steps
            value_source = "the K-step region (+ the plan's amortised share when no epoch began inside it)"
        # SGD reads and writes 3 rows (+ 2 int32 ids); Adam reads and writes w, m, v of 3
        # rows (+ ids + 24 B of per-row step marks); momentum / RMSprop carry one state table
        bytes_per_triple = {"sgd": 24 * d + 8, "adam": 72 * d + 32, "momentum": 48 * d + 32,
                            "rmsprop": 48 * d + 32}[args.optimizer]
        # HBM-side bytes per k_stream launch: NOT measured by this run — replayed from the committed
        # summary of separate rocprofv3 --pmc passes of this same command (profiles/*_pmc_traffic.md:
        # FETCH_SIZE x2 correction + WRITE_SIZE); null when the run is not the profiled configuration
        traffic = None
        tname = TRAFFIC_FILES.get((args.workload, d, args.sampler, args.optimizer))
        tfile = ROOT / "profiles" / (tname or "none")
        if tfile.exists() and args.scale == 1.0 and not batched and not args.item_bias:
            tj = json.loads(tfile.read_text())
            if tj.get("triples_per_launch") == chunk:
                traffic = tj["traffic_bytes_per_launch"]
                if tname == "traffic_r06.json" and (e.lds_launches == 0 or steady is None):
                    traffic = None  # (that file is the LDS-tier kernel on the trained state: not this run)
        achieved = (bytes_per_triple * chunk) / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        out = {
            "metric": "BPR triples/sec at d=128 (1/2/4/8 GPU) + nDCG@100 parity vs reference",
            "value": value,
            "unit": "triples/s",
            "n_gpus": world,
            "steps": steps_behind,  # the steps behind `value` (the driver's K: timed_region.steps)
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "value_source": value_source,
            "early_state": {"value": early_value, "ms_per_step": early_ms, "kernel_ms_avg": early_kernel_ms,
                            "roofline_frac": ((bytes_per_triple * chunk) / (early_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                                              if early_kernel_ms > 0 else None),
                            "note": "the same job on tables a few epochs from their random init (what r1-r5 reported as "
                                    "`value`): the adaptive sampler's negatives are still uniform over the items there"},
            "timed_region": {"steps": args.steps, "warmup": args.warmup, "ms_per_step_measured": dt_measured * 1e3 / args.steps,
                             "value_measured": region_value, "plans_inside": plans_timed,
                             "value_with_amortised_plan": region_value_with_plan},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}-shaped synthetic user-split ({U - 1} users x {I - 1} "
                            f"items, {data.nnz} train triples per GPU), d={d}, {opt_desc} + L2 "
                            f"reg {reg}, {args.sampler} negative sampling"
                            + (f" p={args.adaptive_p}" if args.sampler == "adaptive" else "")
                            + (f", batched STREAM mode (virtual mini-batches of {args.batch_size})" if batched
                               else ", STREAM mode") + f", step = snapshot refresh + {chunk} triples"
                            + ("" if lag == 0.0 else f" (snapshot sorted beside the launch: lag {lag:g}, "
                               f"{split} launch(es) per refresh period, sort masked to {cus} CUs)"),
                "item_bias": bool(args.item_bias),
                "hot_lds": {"rows_asked": lds_rows_asked if not batched else 0, "rows_in_lds_last_launch": e.stream_lds_rows(),
                            "rule": "fast.hot_lds_rows: on while lr x 2 x job triples per launch <= 2,000 (a CU sees the other "
                                    "CUs' updates of these rows one launch late)"},
                "triples_per_step_per_gpu": chunk,
                "refresh_schedule": {"lag": lag, "launches_per_period": split, "launches_sharing_a_snapshot": lsplit,
                                     "side_stream_cus": cus,
                                     "sharded_over_ranks": bool(shard_refresh and not batched)},
                "cadence": (f"{args.cadence}: {ranks_per_period} chunk(s) of {chunk} triples per rank and refresh "
                            f"period (lr x N x chunk = {args.lr * cad_world * chunk:.0f}, N = {cad_world}"
                            f"{' EMULATED on one GPU, collectives left out' if emu else ''}, budget "
                            f"{__import__('revisit_bpr.fast', fromlist=['x']).STALENESS_BUDGET:.0f}); per chunk one "
                            f"snapshot refresh + one cold-row reconciliation, {pieces} launch(es), hot tier of "
                            f"{sync._hb.shape[0] if sync.hot_tier else 0} rows exchanged after every launch"
                            ) if (world > 1 or emu) else "single GPU",
                "steps_per_epoch": n_chunks,
                "plan_epoch": {"mode": ("per chunk, one step ahead, on the side stream behind the sort "
                                        "(bpr_plan_chunk): inside every step") if jit else
                                       ("bpr_plan_epoch per epoch, the next epoch's beside this epoch's launches on a "
                                        "third stream masked to the sorter's CUs") if plan_ahead else "bpr_plan_epoch per epoch",
                               "ms": plan_ms, "inside_timed_region": plans_timed,
                               "amortised_share_added_ms_per_step":
                                   max(0.0, args.steps / n_chunks - plans_timed) * plan_ms / args.steps,
                               "ms_per_step_measured": dt_measured * 1e3 / args.steps},
                "parallelism": f"user-sharded x{world}, item table replicated, async delta "
                               f"all-reduce every {args.sync_every} step(s)" if world > 1 else "single GPU",
                "parity": PARITY_OF_TIMED_SCHEDULE if (world == 1 and lag >= 1.0 and not batched) else None,
                "mean_bpr_loss": float(sc[0] / max(sc[3], 1.0)),
                "triples_counted_by_kernel": int(round(float(sc[3]))),
                "item_table_abs_sum_before_after": [q_before[1], q_after[1]],
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("k_vstream" if batched else "k_stream") + "<G=%d,E=%d,%s>" % (32 if d <= 128 else 64, max(1, -(-d // (32 if d <= 128 else 64))), args.sampler.upper()),
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                # the figure north_star quotes: bytes READ per triple only (3 rows + 2 ids)
                "read_only_frac": ((12 * d + 8) * chunk / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                                   if kernel_ms > 0 and args.optimizer == "sgd" else None),
                "traffic": traffic,
                "traffic_measured_in_this_run": False,
                "traffic_source": (f"REPLAYED, not measured by this run: profiles/{tname} (separate rocprofv3 --pmc FETCH_SIZE / "
                                   "WRITE_SIZE passes of this command on an MI355X, bytes per launch of the same state, "
                                   "gfx950 x2 read correction; rocprofv3 cannot wrap a run from inside it)") if traffic else None,
                "algorithmic_bytes_per_launch": bytes_per_triple * chunk,
                "bytes_per_triple": bytes_per_triple,
                "kernel_ms_avg": kernel_ms,
                "launches": launches,
            },
        }
        if steady is not None:
            out["steady_state"] = steady
        if sus_epochs > 0:
            out["sustained"] = {
                "epochs": sus_epochs, "steps": sus_epochs * n_chunks,
                "ms_per_step": sus_dt * 1e3 / (sus_epochs * n_chunks),
                "value": sus_epochs * n_chunks * chunk * world / sus_dt,
                "note": "whole epochs by wall clock, every bpr_plan_epoch in place (nothing modelled)"}
        if not batched and kernel_ms > 0:
            # what bounds k_stream is the L2 atomic units, not HBM: full-line fp32 atomic requests
            lpt = 2 * lines_per_row + user_atomic_rows * lines_per_row
            rate = lpt * chunk / (kernel_ms * 1e-3)
            out["roofline_atomic"] = {
                "bound": "l2-atomic-units", "unit": "line-atomics/s", "achieved": rate,
                "peak": ATOMIC_LINES_PEAK, "frac": rate / ATOMIC_LINES_PEAK,
                "line_atomics_per_triple": lpt,
                "user_rows_cut_per_triple": user_atomic_rows,
                "peak_source": "profiles/ubench_atomic_r02.txt (tools/ubench/atomic_bench.hip, uniform items)",
            }
        if sync is not None:
            # how the item reconciliation sits next to the step: the all-reduce runs on a side
            # stream under the next step's kernels; it is hidden as long as it is shorter than a step
            st = sync.timing_read()
            st["sync_every_steps"] = args.sync_every
            st["hidden_if_below_ms"] = dt * 1e3 / args.steps * args.sync_every
            out["item_sync"] = st
        if rccl_ranks_seen is not None:
            out["rccl_ranks_seen"] = rccl_ranks_seen
            out["dist_backend"] = dist.get_backend()
            out["forced_distributed"] = forced
        if not args.no_cpu_baseline:
            smp = "uniform" if args.sampler == "given" else args.sampler
            out["cpu_baseline"] = cpu_reference_op_sequence(data, d, reg, args.lr, args.adaptive_p,
                                                            smp, args.seed, optimizer=args.optimizer,
                                                            betas=tuple(args.betas), B=args.batch_size)
            # the C restatement of the same path with OpenMP over triples beside it: sparse
            # apply, no [B, I] weights — kinder to the CPU than the reference's own op sequence (plain SGD
            # step whatever --optimizer: the C port has no sparse form of the stateful optimizers)
            out["cpu_baseline_c_port"] = cpu_baseline(data, d, reg, args.lr, args.adaptive_p,
                                                      args.cpu_seconds, args.seed, smp, B=args.batch_size)
        print(json.dumps(out), flush=True)
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
