"""N>1 path on CPU (gloo, world_size 2): user sharding + replicated item table reconciled by an
all-reduce of deltas (revisit_bpr/distributed.py).  The per-rank training step is played by the CPU
oracle here (tests may use it); on GPUs it is the HIP engine — the protocol is the same."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from revisit_bpr.datasets import synthetic
from revisit_bpr.distributed import ItemSync, balanced_user_shards, owner_of


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_balanced_user_shards():
    data = synthetic.generate(3000, 400, 60000, median_per_user=10, seed=2)
    for world in (1, 2, 4, 8):
        b = balanced_user_shards(data.indptr, world)
        assert b[0] == 0 and b[-1] == data.num_users and np.all(np.diff(b) >= 0)
        loads = np.diff(data.indptr[b])
        assert loads.sum() == data.nnz
        assert loads.max() <= data.nnz / world * 1.1 + np.diff(data.indptr).max()
        own = owner_of(data.users, b)
        assert own.min() >= 0 and own.max() < world
        for r in range(world):
            u = data.users[own == r]
            assert u.size == 0 or (u.min() >= b[r] and u.max() < b[r + 1])


def _worker_sync(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    Q0 = torch.randn(50, 8)
    Q = Q0.clone()
    sync = ItemSync([Q])
    mine = torch.zeros_like(Q)
    mine[rank::world] = float(rank + 1)
    Q += mine
    sync.sync()
    want = Q0 + sum((torch.zeros_like(Q0).index_fill_(0, torch.arange(r, 50, world), float(r + 1))
                     for r in range(world)))
    ok_blocking = torch.allclose(Q, want)
    # asynchronous form: contributions made after start() survive finish()
    sync.start()
    late = torch.zeros_like(Q)
    late[rank] = 100.0 * (rank + 1)
    Q += late
    sync.finish()
    mid = Q.clone()
    sync.sync()
    late_all = torch.zeros_like(Q0)
    for r in range(world):
        late_all[r] = 100.0 * (r + 1)
    ok_async = torch.allclose(mid, want + late) and torch.allclose(Q, want + late_all)
    # DDP-style mean scale
    Q2 = Q0.clone()
    s2 = ItemSync([Q2], scale=1.0 / world)
    Q2 += float(rank + 1)
    s2.sync()
    ok_scale = torch.allclose(Q2, Q0 + sum(range(1, world + 1)) / world)
    out[rank] = (ok_blocking, ok_async, ok_scale)
    dist.destroy_process_group()


def test_item_sync_gloo_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_sync, args=(2, _free_port(), out), nprocs=2, join=True)
    assert dict(out) == {0: (True, True, True), 1: (True, True, True)}


def _worker_train(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = synthetic.generate(600, 150, 9000, median_per_user=10, seed=4)
    d, B, lr, reg = 16, 128, 0.05, (0.001, 0.002, 0.003)
    rng = np.random.default_rng(0)
    P = ((rng.random((data.num_users, d)) - 0.5) / d).astype(np.float32)
    Q = ((rng.random((data.num_items, d)) - 0.5) / d).astype(np.float32)
    P0 = P.copy()
    bounds = balanced_user_shards(data.indptr, world)
    sel = owner_of(data.users, bounds) == rank
    users, items = data.users[sel], data.items[sel]
    tQ = torch.from_numpy(Q)  # shares memory with Q
    sync = ItemSync([tQ])
    for step in range(5):
        u, i = users[step * B:(step + 1) * B], items[step * B:(step + 1) * B]
        neg = oracle.sample_uniform(data.indptr, data.indices, data.num_items, u, seed=1,
                                    offset=(rank << 40) + step * B)
        oracle.step_sgd_sparse(P, Q, None, np.ascontiguousarray(u), np.ascontiguousarray(i), neg, lr, reg)
        sync.sync()
    touched = np.unique(np.nonzero(np.abs(P - P0).sum(1))[0])
    out[rank] = (Q.copy(), touched, bounds)
    dist.destroy_process_group()


def test_user_sharded_training_gloo_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_train, args=(2, _free_port(), out), nprocs=2, join=True)
    (Q0, t0, bounds), (Q1, t1, _) = out[0], out[1]
    assert np.array_equal(Q0, Q1)  # replicas identical after every sync
    assert t0.size and t1.size
    assert t0.max() < bounds[1] <= t1.min()  # each rank only ever touched its own user shard
    # same protocol emulated in one process
    data = synthetic.generate(600, 150, 9000, median_per_user=10, seed=4)
    d, B, lr, reg = 16, 128, 0.05, (0.001, 0.002, 0.003)
    rng = np.random.default_rng(0)
    P = ((rng.random((data.num_users, d)) - 0.5) / d).astype(np.float32)
    Q = ((rng.random((data.num_items, d)) - 0.5) / d).astype(np.float32)
    own = owner_of(data.users, bounds)
    shards = [(data.users[own == r], data.items[own == r]) for r in range(2)]
    for step in range(5):
        deltas = []
        for r, (users, items) in enumerate(shards):
            Pr, Qr = P, Q.copy()  # P rows are disjoint across ranks
            u, i = users[step * B:(step + 1) * B], items[step * B:(step + 1) * B]
            neg = oracle.sample_uniform(data.indptr, data.indices, data.num_items, u, seed=1,
                                        offset=(r << 40) + step * B)
            oracle.step_sgd_sparse(Pr, Qr, None, np.ascontiguousarray(u), np.ascontiguousarray(i), neg, lr, reg)
            deltas.append(Qr - Q)
        Q = Q + deltas[0] + deltas[1]
    assert np.allclose(Q, Q0, atol=1e-6)


def _worker_rounds(rank, world, port, out):
    """Ranks whose shards cut into different numbers of chunks still issue the same sequence of
    collectives (an extra all-reduce on one rank is an RCCL hang)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Q = torch.zeros(20, 4)
    sync = ItemSync([Q])
    chunk, n = 100, (250 if rank == 0 else 390)  # 3 vs 4 chunks
    rounds = sync.max_over_ranks(-(-n // chunk))
    trained = 0
    for k in range(rounds):
        lo, hi = k * chunk, min((k + 1) * chunk, n)
        if lo < hi:
            Q[rank] += float(hi - lo)
            trained += hi - lo
        sync.step()
    sync.finish()
    sync.sync()
    out[rank] = (rounds, trained, Q[0, 0].item(), Q[1, 0].item())
    dist.destroy_process_group()


def test_uneven_shards_keep_collectives_matched():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_rounds, args=(2, _free_port(), out), nprocs=2, join=True)
    assert dict(out) == {0: (4, 250, 250.0, 390.0), 1: (4, 390, 250.0, 390.0)}


def test_local_world_runs_the_item_sync_protocol_in_one_process():
    """distributed.LocalWorld: N ranks of ItemSync in ONE process, the all-reduce resolved when the
    last rank has contributed — the same tables as the protocol's algebra written out (every rank's
    delta of step k reaches the others one step later, the bases stay bit-identical), for the fused
    step() and for start() / finish(); reading a sum before every rank contributed raises."""
    from revisit_bpr.distributed import LocalWorld

    world, I, d, steps = 4, 37, 8, 5
    g = torch.Generator().manual_seed(3)
    Q0 = torch.randn(I, d, generator=g)
    upd = [[torch.randn(I, d, generator=g) * 0.01 for _ in range(world)] for _ in range(steps)]
    lw = LocalWorld(world)
    Qs = [Q0.clone() for _ in range(world)]
    syncs = [ItemSync([Qs[r]], comm=lw.member(r)) for r in range(world)]
    assert all(s.world == world and s.rank == r for r, s in enumerate(syncs))
    for k in range(steps):
        for r in range(world):
            Qs[r] += upd[k][r]
            syncs[r].step()
    for r in range(world):
        syncs[r].finish()
    want = Q0 + sum(sum(u) for u in upd)
    for r in range(world):
        assert torch.allclose(Qs[r], want, atol=1e-5)
        assert torch.equal(syncs[r].base[0], syncs[0].base[0])
    # one step late: after step k (before finish) rank r holds its own updates up to k and the
    # others' up to k - 1
    lw = LocalWorld(2)
    Qs = [Q0.clone() for _ in range(2)]
    syncs = [ItemSync([Qs[r]], comm=lw.member(r)) for r in range(2)]
    for k in range(2):
        for r in range(2):
            Qs[r] += upd[k][r]
            syncs[r].step()
    assert torch.allclose(Qs[0], Q0 + upd[0][0] + upd[1][0] + upd[0][1], atol=1e-6)
    assert torch.allclose(Qs[1], Q0 + upd[0][1] + upd[1][1] + upd[0][0], atol=1e-6)
    # a rank that reads before the others contributed
    lw = LocalWorld(2)
    a = ItemSync([Q0.clone()], comm=lw.member(0))
    ItemSync([Q0.clone()], comm=lw.member(1))
    a.start()
    with pytest.raises(RuntimeError, match="round-robin"):
        a.finish()


def test_staleness_budget_and_schedule_rules():
    """fast.launches_per_period: lr x world x chunk <= STALENESS_BUDGET — a full
    period per rank at the benchmark config's lr 0.001 for up to 8 ranks, 1 / 2 / 4 chunks at the
    reference's tuned lr 0.0094, period / 2.5 N at lr 0.05 (capped at 4 N); fast.auto_schedule: the
    overlapped snapshot schedule on 32 CUs for the ML-20M shape (binned sort), on 96 for MSD d = 256."""
    from revisit_bpr import fast

    period = 199_168
    assert [fast.launches_per_period(0.001, w, period) for w in (1, 2, 4, 8)] == [1, 1, 1, 1]
    assert [fast.launches_per_period(0.0094, w, period) for w in (1, 2, 4, 8)] == [1, 1, 2, 4]
    assert [fast.launches_per_period(0.05, w, period) for w in (1, 2, 4, 8)] == [1, 5, 10, 20]
    assert fast.launches_per_period(1.0, 8, period) == 4 * 8
    for lr in (0.001, 0.0094, 0.05):
        for w in (2, 4, 8):
            k = fast.launches_per_period(lr, w, period)
            assert k == 4 * w or lr * w * (period / k) <= fast.STALENESS_BUDGET
    assert fast.auto_schedule(20109, 128, period) == (1.0, 32)  # (64 until the binned sort of r5)
    assert fast.auto_schedule(41141, 256, 436_992) == (1.0, 64)  # (96 until the split binned sort of r5)
    lag, cus = fast.auto_schedule(4800, 64, 40_704)
    assert lag == 1.0 and cus == 32
    # r5: a lagged snapshot misses up to two launches of updates and is held to a budget of its own —
    # lr x 2 x launch <= LAG_BUDGET; r6 moved it from 4,000 to 2,000: lr 0.01 at the ML-20M period (3,983) leads exact
    # mini-batches by +0.0022 / +0.0044 nDCG@100 at epochs 6 / 8 with 8 seeds (profiles/r06_parity_study.md)
    assert fast.LAG_BUDGET == 2000.0
    assert fast.lag_within_budget(0.001, period) and fast.lag_within_budget(0.005, period)
    assert not fast.lag_within_budget(0.01, period) and not fast.lag_within_budget(0.05, period)
    assert fast.auto_schedule(20109, 128, period, lr=0.001) == (1.0, 32)
    assert fast.auto_schedule(20109, 128, period, lr=0.01) == (0.0, 0) == fast.auto_schedule(20109, 128, period, lr=0.05)
    assert fast.auto_schedule(4800, 64, 40_704, lr=0.05)[0] == 0.0   # Netflix at lr 0.05: 4,070 > 2,000
    assert fast.auto_schedule(4800, 64, 40_704, lr=0.02)[0] == 1.0
    # the LDS tier of the hot block (r6) goes by the same rule, counted in triples of the whole job
    assert fast.hot_lds_rows(0.001, period) == fast.HOT_LDS_ROWS and fast.hot_lds_rows(0.001, period, world=8) > 0
    assert fast.hot_lds_rows(0.01, period) == 0 and fast.hot_lds_rows(0.001, period, world=32) == 0
    # the asynchronous cut only where the sorter has slack on its masked CUs: the one-workgroup binned sort's tables
    assert fast.auto_async_cut(20109, 32) and not fast.auto_async_cut(20109, 0)
    assert not fast.auto_async_cut(41141, 64) and not fast.auto_async_cut(92090, 128)
