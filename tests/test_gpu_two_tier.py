"""Two-tier item reconciliation: the hot rows' deltas travel after
every (sub-)launch through `bpr_hot_exchange`, the cold rows once per period through
`bpr_item_fold_delta`.  Several ranks run in ONE process on one GPU over `distributed.LocalWorld`
(the product ItemSync with its collective resolved in-process: same data flow as N processes).

The deterministic test holds the product to a dense restatement of the protocol written with plain
torch tensors (every rank's launch on its own plain engine, deltas taken as table differences, every
sum folded exactly one protocol step after it was cut); the trainer test runs `fast.StreamTrainer`
with uneven shards — a rank that runs out of triples must still enter every collective of a round.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _problem(U, I, d, n, seed):
    rng = np.random.default_rng(seed)
    P = rng.normal(0, 0.2, (U, d)).astype(np.float32)
    Q = rng.normal(0, 0.2, (I, d)).astype(np.float32)
    P[0] = 0
    Q[0] = 0
    lens = rng.integers(1, 12, U)
    lens[0] = 0
    rows = [np.sort(rng.choice(np.arange(1, I), size=int(k), replace=False)).astype(np.int32) for k in lens]
    indptr = np.zeros(U + 1, np.int64)
    indptr[1:] = np.cumsum(lens)
    users = np.sort(rng.integers(1, U, n)).astype(np.int32)
    pos = (1 + (rng.zipf(1.4, n) % (I - 1))).astype(np.int32)  # skewed: the hot rows carry most updates
    return P, Q, indptr, np.concatenate(rows), users, pos


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("world,hot_split,H,d,lds", [(2, 1, 16, 64, 0), (3, 2, 40, 128, 0), (4, 3, 8, 256, 0),
                                                     (2, 1, 16, 64, 9), (3, 2, 40, 128, 24), (4, 3, 12, 256, 12)])
def test_two_tier_equals_the_dense_protocol(world, hot_split, H, d, fused, lds):
    """lds > 0 (r6): the launches run the LDS-tier kernel (`bpr_set_hot_lds`, forced for these small launches) — the
    `lds` hottest rows of the tier's H take a workgroup's updates in LDS and are flushed into the block the hot tier
    exchanges, the rest of the H go through the global block as before: the same protocol, the same tables.
    fused: the last launch of a round leaves its epilogue to ONE pass — hot-tier step + cold step +
    the cut of the next snapshot's keys (bpr_sync_cut, ItemSync.step_cut) — instead of three kernels;
    same tables, and the snapshot sorted from that cut is the oracle's order of the table."""
    from revisit_bpr import engine as eng
    from revisit_bpr.distributed import ItemSync, LocalWorld

    U, I, n_round, rounds = 240, 150, 360, 3
    P0, Q0, indptr, indices, users, pos = _problem(U, I, d, world * n_round * rounds, seed=world + d)
    dev = torch.device("cuda")
    reg, lr = (0.01, 0.02, 0.03), 0.05
    counts = torch.bincount(torch.from_numpy(pos).long(), minlength=I)

    def engine(lds_rows=0):
        e = eng.Engine(torch.from_numpy(P0).to(dev), torch.from_numpy(Q0).to(dev))
        e.set_reg(*reg)
        e.set_optimizer(eng.OPT_SGD, lr=lr)
        e.bind_seen_csr(torch.from_numpy(indptr).to(dev), torch.from_numpy(indices).to(dev))
        e.set_stream_opts(True, 0)
        e.set_hot_lds(lds_rows, always=True)
        return e

    u_d, p_d = torch.from_numpy(users).to(dev), torch.from_numpy(pos).to(dev)

    def piece(r, k, p):  # rank r's triples of round k, piece p
        lo = (k * world + r) * n_round
        a, b = lo + n_round * p // hot_split, lo + n_round * (p + 1) // hot_split
        return a, b

    def launch(e, r, k, p, cut=False):
        a, b = piece(r, k, p)
        e.train_stream(u_d[a:b], p_d[a:b], sampler=eng.NEG_UNIFORM, seed=7, offset=(r << 40) + a,
                       max_inflight=1, cut=cut)

    # ---- the product: ItemSync with the hot tier over LocalWorld
    lw = LocalWorld(world)
    es = [engine(lds) for _ in range(world)]
    syncs = [ItemSync([es[r].Q], comm=lw.member(r), engine=es[r], hot_rows=H, item_counts=counts)
             for r in range(world)]
    assert all(s.hot_tier for s in syncs) and es[0].hot_rows() == H
    hot = syncs[0].hot_items.numpy()
    import oracle

    for k in range(rounds):
        for p in range(hot_split):
            for r in range(world):
                last = fused and p == hot_split - 1
                launch(es[r], r, k, p, cut=last)
                if last:
                    syncs[r].step_cut()
                else:
                    syncs[r].hot_step()
        for r in range(world):
            if not fused:
                syncs[r].step()
        if fused:  # the cut rode on the fused pass: begin only queues the sort
            q_now = es[0].Q.cpu().numpy()
            es[0].adaptive_refresh_begin()
            es[0].adaptive_refresh_commit()
            QT, _ = oracle.adaptive_stats(q_now)
            assert np.array_equal(es[0].adaptive_snapshot()[0].cpu().numpy(), oracle.adaptive_order(QT)), k
    for r in range(world):
        syncs[r].hot_finish()
        syncs[r].finish()
    torch.cuda.synchronize()
    assert all((e.lds_launches > 0) == (lds > 0) for e in es) and (lds == 0 or es[0].stream_lds_rows() == min(lds, H))

    # ---- the protocol restated densely: plain engines, deltas as table differences
    ps = [engine() for _ in range(world)]
    is_hot = np.zeros(I, bool)
    is_hot[hot] = True
    hot_t = torch.from_numpy(is_hot).to(dev)
    hot_in_flight = None   # per rank: the hot-row deltas cut after the previous piece
    cold_in_flight = None  # per rank: the cold-row deltas cut after the previous round
    cold_acc = [torch.zeros(I, d, device=dev) for _ in range(world)]
    for k in range(rounds):
        for p in range(hot_split):
            cut = []
            for r in range(world):
                before = ps[r].Q.clone()
                launch(ps[r], r, k, p)
                dl = ps[r].Q - before
                cut.append(dl * hot_t[:, None])
                cold_acc[r] += dl * (~hot_t)[:, None]
            if hot_in_flight is not None:  # folded one exchange late: everybody else's hot deltas
                for r in range(world):
                    ps[r].Q.add_(sum(hot_in_flight[s] for s in range(world) if s != r))
            hot_in_flight = cut
        if cold_in_flight is not None:
            for r in range(world):
                ps[r].Q.add_(sum(cold_in_flight[s] for s in range(world) if s != r))
        cold_in_flight = [c.clone() for c in cold_acc]
        for c in cold_acc:
            c.zero_()
    for r in range(world):
        ps[r].Q.add_(sum(hot_in_flight[s] for s in range(world) if s != r))
        ps[r].Q.add_(sum(cold_in_flight[s] for s in range(world) if s != r))
    torch.cuda.synchronize()
    for r in range(world):
        got, want = es[r].Q.cpu().numpy(), ps[r].Q.cpu().numpy()
        err = np.abs(got - want).max()
        assert err <= 5e-6, (r, err)
        assert np.abs(es[r].P.cpu().numpy() - ps[r].P.cpu().numpy()).max() <= 5e-6
    # the reconciled state: every replica is the same table, and the hot base is bit-identical
    for r in range(1, world):
        assert (es[r].Q - es[0].Q).abs().max().item() <= 2e-6
        assert torch.equal(syncs[r]._hb, syncs[0]._hb)
        assert torch.equal(syncs[r].base[0][~hot_t], syncs[0].base[0][~hot_t])
    # and it moved: hot rows by far the most
    moved = (es[0].Q - torch.from_numpy(Q0).to(dev)).abs().sum(1).cpu().numpy()
    assert moved[hot].mean() > 1.5 * moved[~is_hot].mean() > 0
    for s in syncs:
        s.close()


@pytest.mark.parametrize("cadence,hot_rows,hot_split,lag", [("rank", 64, 2, 1), ("rank", 32, 1, 0),
                                                             ("job", 0, 1, 0)])
def test_trainers_over_local_world_with_uneven_shards(golden_dir, cadence, hot_rows, hot_split, lag):
    """fast.StreamTrainer on 3 ranks of very different size (the smallest runs out of triples rounds
    before the largest): every round's exchanges line up (LocalWorld raises otherwise), replicas
    agree after the epoch, and the job learns like one rank does."""
    from revisit_bpr.distributed import ItemSync, LocalWorld
    from revisit_bpr.evaluation import evaluate_topk
    from revisit_bpr.fast import StreamTrainer
    from revisit_bpr.models import BPR
    from revisit_bpr.models.bpr import MF

    d = np.load(golden_dir / "e2e_data.npz")
    dev = torch.device("cuda")
    U, I = int(d["num_users"]), int(d["num_items"])
    t = {k: torch.from_numpy(d[k]).to(dev) for k in ("users", "items", "indptr", "indices", "eval_users",
                                                     "eval_indptr", "eval_items")}
    bounds = np.array([0, U // 10, U // 3, U])  # 3 shards, ~1 : 2.5 : 7 by users
    own = np.searchsorted(bounds, d["users"], side="right") - 1
    world = 3
    counts = torch.bincount(t["items"].long(), minlength=I)

    def job(world):
        lw = LocalWorld(world)
        every = max(1, int(I * np.log(I) / 256))
        n_r = [int((own == r).sum()) if world > 1 else len(own) for r in range(world)]
        chunk = [max(1, min(every * 256 // (world if cadence == "job" else 1), n)) for n in n_r]
        rounds = max(-(-n // c) for n, c in zip(n_r, chunk))
        trs, feats = [], []
        for r in range(world):
            torch.manual_seed(13)
            model = BPR(fuse_forward=True, reg_alphas={"user": 0.0016, "item": 0.0001, "neg": 0.00375},
                        logits_model=MF(torch.nn.Embedding(U, 32, padding_idx=0),
                                        torch.nn.Embedding(I, 32, padding_idx=0))).to(dev)
            f = model.logits_model.get_features()
            mine = torch.from_numpy(own == r).to(dev) if world > 1 else torch.ones(len(own), dtype=torch.bool, device=dev)
            sync = ItemSync([f["item"].data], comm=lw.member(r), engine=model.engine(), hot_rows=hot_rows,
                            item_counts=counts) if world > 1 else None
            trs.append(StreamTrainer(model, t["users"][mine].contiguous(), t["items"][mine].contiguous(),
                                     t["indptr"], t["indices"], lr=0.05, sampler="adaptive", adaptive_p=0.05,
                                     seed=3, rank=r, item_sync=sync, world=world, cadence=cadence,
                                     hot_split=hot_split, rounds=rounds,
                                     **({"refresh_lag": 1.0, "refresh_cus": 64} if lag else {})))
            feats.append(f)
        for _ in range(6):
            for tr in trs:
                tr.epoch_begin()
            gens = [tr.epoch_iter() for tr in trs]
            alive = True
            while alive:
                alive = False
                for tr, g in zip(trs, gens):
                    with tr.stream_scope():
                        try:
                            next(g)
                            alive = True
                        except StopIteration:
                            pass
            stats = [tr.epoch_end() for tr in trs]
        assert sum(s["triples"] for s in stats) == len(own)
        P = feats[0]["user"].data.clone()
        for r in range(1, world):
            P[int(bounds[r]):int(bounds[r + 1])] = feats[r]["user"].data[int(bounds[r]):int(bounds[r + 1])]
        for r in range(1, world):
            assert (feats[r]["item"].data - feats[0]["item"].data).abs().max().item() < 1e-5
        m = evaluate_topk(P, feats[0]["item"].data, None, t["eval_users"], t["eval_indptr"], t["eval_items"],
                          t["indptr"], t["indices"], ks=(100,))
        for tr in trs:
            if tr.item_sync is not None:
                tr.item_sync.close()
        return m["ndcg@100"]

    one, many = job(1), job(world)
    assert one > 0.25 and abs(many - one) < 0.03, (one, many)


def test_cut_launch_under_the_hot_tier_keeps_its_statistics_without_sync_cut():
    """`train_stream(cut=True)` under the hot tier leaves its loss partials to the
    `bpr_sync_cut` that normally follows.  On the other documented two-tier path — hot exchange + cold
    step as separate calls — nobody calls it: the sums must arrive anyway (flushed by the exchange, or by
    the next launch before it overwrites the partials), once, and nothing stays pending."""
    from revisit_bpr import engine as eng
    from revisit_bpr.distributed import ItemSync, LocalWorld

    U, I, d, n = 240, 150, 64, 900
    P0, Q0, indptr, indices, users, pos = _problem(U, I, d, 3 * n, seed=5)
    dev = torch.device("cuda")
    counts = torch.bincount(torch.from_numpy(pos).long(), minlength=I)
    u_d, p_d = torch.from_numpy(users).to(dev), torch.from_numpy(pos).to(dev)
    totals = []
    for path in ("sync_cut", "hot_step", "next_launch"):
        e = eng.Engine(torch.from_numpy(P0).to(dev), torch.from_numpy(Q0).to(dev))
        e.set_reg(0.01, 0.02, 0.03)
        e.set_optimizer(eng.OPT_SGD, lr=0.05)
        e.bind_seen_csr(torch.from_numpy(indptr).to(dev), torch.from_numpy(indices).to(dev))
        e.set_stream_opts(True, 0)
        sync = ItemSync([e.Q], comm=LocalWorld(1).member(0), engine=e, hot_rows=16, item_counts=counts,
                        force_tiers=True)
        assert sync.hot_tier
        sc = torch.zeros(4, device=dev)
        for k in range(3):
            e.train_stream(u_d[k * n:(k + 1) * n], p_d[k * n:(k + 1) * n], sampler=eng.NEG_UNIFORM, seed=7,
                           offset=k * n, max_inflight=1, scalars=sc, cut=True)
            if path == "sync_cut":
                sync.step_cut()
            elif path == "hot_step":
                sync.hot_step()
                sync.step()
            # "next_launch": nothing in between — the next launch must not overwrite unsummed partials
        sync.hot_finish()
        sync.finish()
        sync.close()
        torch.cuda.synchronize()
        assert int(sc[3]) == 3 * n, (path, sc)
        totals.append(sc.cpu().numpy())
    # the same triples with the same negatives (max_inflight = 1: sequential): the same sums
    assert np.allclose(totals[0], totals[1], rtol=1e-5) and np.allclose(totals[0], totals[2], rtol=1e-4)


def test_two_ranks_with_an_item_bias_keep_the_reconciled_vector(monkeypatch):
    """`StreamTrainer.epoch_begin` switched the launches' bias tracking on unconditionally — but
    with several ranks ItemSync folds the other ranks' item_bias deltas into the vector BETWEEN launches (through
    ctx-less entry points), so the next launch trained on a stale one-item-per-line copy and its epilogue wrote that
    copy back over the reconciled vector.  Two in-process ranks, the model of the RQ configs (item_bias on),
    ItemSync([item, item_bias]); one group in flight per launch so that a run is deterministic: the product must
    equal the same job with tracking forced off (every launch refills), the replicas must agree, the bias must
    have moved on both."""
    from revisit_bpr import engine as eng
    from revisit_bpr.distributed import ItemSync, LocalWorld
    from revisit_bpr.fast import StreamTrainer
    from revisit_bpr.models import BPR
    from revisit_bpr.models.bpr import MF

    U, I, d, n = 300, 120, 64, 2400
    _, _, indptr, indices, users, pos = _problem(U, I, d, n, seed=11)
    dev = torch.device("cuda")
    own = (users >= U // 2).astype(np.int64)
    u_d, p_d = torch.from_numpy(users).to(dev), torch.from_numpy(pos).to(dev)

    def job(force_off):
        if force_off:
            real = eng.Engine.set_bias_tracking
            monkeypatch.setattr(eng.Engine, "set_bias_tracking", lambda self, on: real(self, False))
        lw = LocalWorld(2)
        trs, feats = [], []
        every = max(1, int(I * np.log(I) / 64))
        chunk = max(1, every * 64 // 2)  # cadence "job": a period of the job = two rank chunks
        rounds = max(-(-int((own == r).sum()) // chunk) for r in range(2))  # (LocalWorld has no blocking collectives)
        for r in range(2):
            torch.manual_seed(13)
            model = BPR(fuse_forward=True, reg_alphas={"user": 0.0025, "item": 0.0025, "neg": 0.00025},
                        logits_model=MF(torch.nn.Embedding(U, d, padding_idx=0), torch.nn.Embedding(I, d, padding_idx=0),
                                        item_bias=True)).to(dev)
            f = model.logits_model.get_features()
            mine = torch.from_numpy(own == r).to(dev)
            sync = ItemSync([f["item"].data, f["item_bias"].data], comm=lw.member(r), engine=model.engine())
            trs.append(StreamTrainer(model, u_d[mine].contiguous(), p_d[mine].contiguous(), torch.from_numpy(indptr).to(dev),
                                     torch.from_numpy(indices).to(dev), lr=0.05, sampler="uniform", batch_size=64, seed=3,
                                     rank=r, item_sync=sync, world=2, max_inflight=1, rounds=rounds))
            feats.append(f)
        assert trs[0].chunk == chunk and trs[0].rounds > 2  # several reconciliations inside an epoch
        for _ in range(3):
            for tr in trs:
                tr.epoch_begin()
            gens = [tr.epoch_iter() for tr in trs]
            alive = True
            while alive:
                alive = False
                for tr, g in zip(trs, gens):
                    try:
                        next(g)
                        alive = True
                    except StopIteration:
                        pass
            for tr in trs:
                tr.epoch_end()
        out = [(f["item"].data.clone(), f["item_bias"].data.clone()) for f in feats]
        for tr in trs:
            tr.item_sync.close()
        monkeypatch.undo()
        return out

    got, want = job(False), job(True)
    for r in range(2):
        assert got[r][1].abs().max().item() > 1e-3  # the bias learned something
    assert (got[0][1] - got[1][1]).abs().max().item() < 1e-6 and (got[0][0] - got[1][0]).abs().max().item() < 1e-5
    assert (got[0][1] - want[0][1]).abs().max().item() < 1e-5, (got[0][1] - want[0][1]).abs().max().item()
    assert (got[0][0] - want[0][0]).abs().max().item() < 1e-5
