"""GPU tests of the host-side mirror of the reference API: BPR / MF with a stock torch.optim
optimizer, the samplers, the trainer — all driving the HIP engine — against (a) the dense PyTorch
restatement on the same device (`set_backend("torch")`, the "PyTorch-ROCm ref" of BASELINE config 2)
and (b) the reference's golden vectors."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def build(U, I, d, reg, item_bias=False, seed=0):
    from revisit_bpr.models import BPR
    from revisit_bpr.models.bpr import MF

    torch.manual_seed(seed)
    model = BPR(fuse_forward=True, reg_alphas=reg,
                logits_model=MF(torch.nn.Embedding(U, d, padding_idx=0),
                                torch.nn.Embedding(I, d, padding_idx=0), item_bias=item_bias))
    return model.cuda()


def batches(U, I, B, steps, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(steps):
        u = torch.randint(1, U, (B,), generator=g)
        i = torch.randint(1, I, (B, 1), generator=g)
        j = torch.randint(1, I, (B, 1), generator=g)
        u[1] = u[0]
        i[3] = i[2]
        j[5] = i[4]
        out.append({"user": u.cuda(), "item": i.cuda(), "neg": j.cuda()})
    return out


OPTS = {
    "sgd": lambda p: torch.optim.SGD(p, lr=0.05),
    "nesterov": lambda p: torch.optim.SGD(p, lr=0.05, momentum=0.9, nesterov=True),
    "adam": lambda p: torch.optim.Adam(p, lr=0.01, betas=(0.9, 0.999)),
    "adam01": lambda p: torch.optim.Adam(p, lr=0.01, betas=(0.1, 0.999)),
    "rmsprop": lambda p: torch.optim.RMSprop(p, lr=0.01, alpha=0.9),
    # momentum is in the Optuna space of configs/RQ2/optimizers/rmsprop-ml-20m.yaml.j2:62-64
    "rmsprop_mom": lambda p: torch.optim.RMSprop(p, lr=0.005, alpha=0.9, momentum=0.8),
}


@pytest.mark.parametrize("opt_name", list(OPTS))
@pytest.mark.parametrize("item_bias", [False, True])
def test_reference_loop_fused_equals_dense_torch(opt_name, item_bias):
    """model(batch) / loss.backward() / optimizer.step() / zero_grad() — the reference's loop — gives
    the same parameters through the HIP engine as through dense autograd + torch.optim."""
    from revisit_bpr.models.bpr import set_backend

    U, I, d, B, steps = 300, 200, 64, 128, 6
    reg = {"user": 0.0016, "item": 0.0001, "neg": 0.00375}
    data = batches(U, I, B, steps, seed=3)
    results = {}
    for backend in ("hip", "torch"):
        set_backend(backend)
        try:
            model = build(U, I, d, reg, item_bias, seed=11)
            if item_bias:
                with torch.no_grad():
                    model.logits_model._item_bias.copy_(torch.linspace(-0.1, 0.1, I))
            opt = OPTS[opt_name](model.parameters())
            model.train()
            losses = []
            for b in data:
                out = model(b)
                assert set(out) >= {"logits_pos", "logits_neg", "logits", "bpr_loss", "l2_reg", "loss"}
                assert out["logits"].shape == (B, 1)
                out["loss"].backward()
                opt.step()
                opt.zero_grad()
                losses.append(float(out["loss"].detach()))
            model.eval()  # fused: brings lazily-updated rows up to date
            results[backend] = ({k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                                losses)
            if backend == "hip":
                assert all(p.grad is None for p in model.parameters())
        finally:
            set_backend("hip")
    (sd_h, l_h), (sd_t, l_t) = results["hip"], results["torch"]
    assert np.allclose(l_h, l_t, rtol=2e-5)
    # RMSprop divides by sqrt(v) ~ |g|: elements whose gradient is ~0 amplify fp32 rounding of g
    atol = 1e-4 if opt_name.startswith("rmsprop") else 2e-5
    for k in sd_t:
        assert torch.allclose(sd_h[k], sd_t[k], rtol=0, atol=atol), (k, (sd_h[k] - sd_t[k]).abs().max())


def test_fused_against_reference_golden(golden_dir):
    """Same loop on the reference's own fixture (tables from the reference's init, torch.optim.Adam)."""
    from revisit_bpr.models import BPR
    from revisit_bpr.models.bpr import MF

    g = np.load(golden_dir / "math_13_uin_bias.npz")
    U, d = g["P0"].shape
    I = g["Q0"].shape[0]
    model = BPR(fuse_forward=True, reg_alphas={"user": 0.0016, "item": 0.0001, "neg": 0.00375},
                logits_model=MF(torch.nn.Embedding(U, d, padding_idx=0),
                                torch.nn.Embedding(I, d, padding_idx=0), item_bias=True))
    with torch.no_grad():
        model.logits_model._user_emb.weight.copy_(torch.from_numpy(g["P0"]))
        model.logits_model._item_emb.weight.copy_(torch.from_numpy(g["Q0"]))
        model.logits_model._item_bias.copy_(torch.from_numpy(g["b0"]))
    model = model.cuda().train()
    opt = torch.optim.Adam(model.parameters(), lr=0.01, betas=(0.9, 0.999))
    for s in range(5):
        b = {"user": torch.from_numpy(g[f"users{s}"]).cuda(),
             "item": torch.from_numpy(g[f"pos{s}"]).unsqueeze(-1).cuda(),
             "neg": torch.from_numpy(g[f"neg{s}"]).unsqueeze(-1).cuda()}
        out = model(b)
        out["loss"].backward()
        opt.step()
        opt.zero_grad()
        assert abs(float(out["loss"]) - float(g[f"adam_09_loss{s + 1}"])) < 1e-4
    feats = model.eval().logits_model.get_features()
    assert np.allclose(feats["user"].detach().cpu().numpy(), g["adam_09_P5"], atol=1e-5)
    assert np.allclose(feats["item"].detach().cpu().numpy(), g["adam_09_Q5"], atol=1e-5)
    assert np.allclose(feats["item_bias"].detach().cpu().numpy(), g["adam_09_b5"], atol=1e-5)
    # the optimizer's state tensors are the live ones (checkpointable)
    st = opt.state[model.logits_model._item_emb.weight]
    assert st["exp_avg"].abs().sum() > 0 and st["exp_avg_sq"].abs().sum() > 0


def test_no_cpu_fallback_in_training():
    from revisit_bpr.models import BPR
    from revisit_bpr.models.bpr import MF

    model = BPR(MF(torch.nn.Embedding(10, 8, padding_idx=0), torch.nn.Embedding(10, 8, padding_idx=0)))
    model.train()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model({"user": torch.tensor([1]), "item": torch.tensor([[1]]), "neg": torch.tensor([[2]])})


def _seen_batch(U, I, B, S, seed):
    g = torch.Generator().manual_seed(seed)
    users = torch.randint(1, U, (B,), generator=g)
    rows = {}
    seen = torch.zeros(B, S, dtype=torch.long)
    for r, u in enumerate(users.tolist()):
        if u not in rows:
            n = int(torch.randint(0, S + 1, (1,), generator=g))
            rows[u] = torch.randperm(I - 1, generator=g)[:n] + 1
        seen[r, :len(rows[u])] = rows[u]
    return users, seen


def test_uniform_sampler_api():
    from revisit_bpr.modules import UniformSampler

    U, I, B, S = 500, 120, 4096, 40
    users, seen = _seen_batch(U, I, B, S, seed=1)
    batch = {"user": users.cuda(), "item": torch.ones(B, 1, dtype=torch.long).cuda(),
             "seen_items": seen.cuda()}
    sampler = UniformSampler(I, torch.Generator(device="cuda").manual_seed(13))
    neg = sampler.sample(batch)
    assert neg.shape == (B, 1) and neg.dtype == torch.long and neg.is_cuda
    n = neg.cpu().squeeze(-1)
    assert n.min() >= 1 and n.max() < I
    assert not (seen == n.unsqueeze(-1)).any()
    neg2 = sampler.sample(batch)
    assert not torch.equal(neg, neg2)  # the stream advances
    # distribution: uniform over unseen for one user (chi-square)
    u0 = int(users[0])
    mask_rows = users == u0
    allowed = torch.ones(I, dtype=torch.bool)
    allowed[0] = False
    allowed[seen[mask_rows][0]] = False
    big = {"user": torch.full((60000,), u0).cuda(), "item": torch.ones(60000, 1, dtype=torch.long).cuda(),
           "seen_items": seen[mask_rows][:1].expand(60000, -1).cuda()}
    cnt = torch.bincount(sampler.sample(big).cpu().squeeze(-1), minlength=I).double()
    assert cnt[~allowed].sum() == 0
    k = int(allowed.sum())
    chi2 = float(((cnt[allowed] - 60000 / k) ** 2 / (60000 / k)).sum())
    assert chi2 < (k - 1) + 5 * math.sqrt(2 * (k - 1)), chi2


def test_adaptive_sampler_api_and_refresh_period():
    from revisit_bpr.modules import AdaptiveSampler

    U, I, d, B, S = 400, 300, 32, 512, 30
    model = build(U, I, d, None, seed=5)
    users, seen = _seen_batch(U, I, B, S, seed=2)
    batch = {"user": users.cuda(), "item": torch.ones(B, 1, dtype=torch.long).cuda(),
             "seen_items": seen.cuda()}
    sampler = AdaptiveSampler(model, I, sampling_prob=0.05,
                              neg_gen=torch.Generator(device="cuda").manual_seed(7), every=3)
    sampler.update_stats()
    order0, _ = model.engine().adaptive_snapshot()
    for it in range(1, 4):
        neg = sampler.sample(batch)
        assert neg.shape == (B, 1) and neg.dtype == torch.long
        n = neg.cpu().squeeze(-1)
        assert n.min() >= 1 and n.max() < I and not (seen == n.unsqueeze(-1)).any()
        with torch.no_grad():  # move the item table so a refresh is visible
            model.logits_model._item_emb.weight[1:].add_(torch.randn(I - 1, d, device="cuda") * 0.01)
        order, _ = model.engine().adaptive_snapshot()
        assert torch.equal(order, order0) == (it < 3)  # refreshed after the 3rd call
    # picks concentrate on the extremes of the chosen factor: rank << I/2 on average
    order, sigma = model.engine().adaptive_snapshot()
    assert sigma.min() > 0


def test_trainer_drives_the_engine():
    from experiments.trainer import ModelEvents, NullAccelerator, Trainer
    from revisit_bpr.modules import UniformSampler

    try:
        from ignite.engine import Events
    except ImportError:
        from experiments.engine_lite import Events

    U, I, d, B, S = 200, 150, 32, 64, 20
    model = build(U, I, d, {"all": 0.001}, seed=3)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    trainer = Trainer(model, opt, NullAccelerator(torch.device("cuda")))
    sampler = UniformSampler(I, torch.Generator(device="cuda").manual_seed(1))
    loader = []
    for k in range(10):
        users, seen = _seen_batch(U, I, B, S, seed=100 + k)
        loader.append({"user": users.cuda(), "item": torch.randint(1, I, (B,)).cuda(),
                       "seen_items": seen.cuda()})
    seen_events = []

    def add_negatives(engine):
        batch = engine.state.batch
        if batch["item"].dim() < 2:
            batch["item"] = batch["item"].unsqueeze(-1)
        batch["neg"] = sampler.sample(batch)

    trainer.add_event("train", Events.GET_BATCH_COMPLETED, add_negatives)
    trainer.add_event("train", ModelEvents.OPTIMIZER_COMPLETED,
                      lambda e: seen_events.append(e.state.optimizer_iteration))
    before = model.logits_model._user_emb.weight.detach().clone()
    state = trainer.run({"train": loader}, epochs=3)
    assert seen_events == list(range(1, 31))
    assert state.iteration == 30 and float(state.metrics["loss"]) > 0
    first, last = float(trainer.engines["train"].state.metrics["loss"]), None
    assert not torch.equal(before, model.logits_model._user_emb.weight)
    assert first < B * math.log(2) * 1.05  # mean batch loss below the untrained value ~B·ln2


def test_evaluate_topk_equals_metric_classes():
    from revisit_bpr.datasets import synthetic
    from revisit_bpr.evaluation import evaluate, evaluate_topk
    from revisit_bpr.metrics import NDCG, Precision, Recall

    data = synthetic.generate_latent(700, 260, 15000, seed=8)
    g = torch.Generator().manual_seed(0)
    P = torch.randn(data.num_users, 32, generator=g).cuda()
    Q = torch.randn(data.num_items, 32, generator=g).cuda()
    b = torch.randn(data.num_items, generator=g).cuda()
    t = {k: torch.from_numpy(getattr(data, k)).cuda()
         for k in ("eval_users", "eval_indptr", "eval_items", "indptr", "indices")}
    ks = (5, 10, 20, 50, 100)
    metrics = {}
    for k in ks:
        metrics[f"ndcg@{k}"], metrics[f"recall@{k}"], metrics[f"precision@{k}"] = \
            NDCG(topk=k), Recall(topk=k), Precision(topk=k)
    args = (P, Q, b, t["eval_users"], t["eval_indptr"], t["eval_items"], t["indptr"], t["indices"])
    slow = evaluate(*args, metrics, block=128)
    fast = evaluate_topk(*args, ks=ks, block=300, auc=True)
    for name, v in slow.items():
        assert abs(fast[name] - v) < 2e-6, (name, fast[name], v)
    # AUC by rank sums == the reference's dense all-pairs definition (metrics/auc.py RocAucMany);
    # on users with held-out items only (a row without positives is 0 / 0 in either form, and the
    # reference's test files list no such user)
    from revisit_bpr.metrics import RocAucMany

    cnt = np.diff(data.eval_indptr)
    keep = cnt > 0
    ptr = np.concatenate([[0], np.cumsum(cnt[keep])]).astype(np.int64)
    args_pos = (P, Q, b, torch.from_numpy(data.eval_users[keep]).cuda(), torch.from_numpy(ptr).cuda(),
                t["eval_items"], t["indptr"], t["indices"])
    dense = evaluate(*args_pos, {"auc": RocAucMany()}, block=64)["auc"]
    ranked = evaluate_topk(*args_pos, ks=(5,), block=300, auc=True)["auc"]
    assert 0.0 < dense < 1.0 and abs(ranked - dense) < 2e-6, (ranked, dense)


def test_user_bias_is_carried_but_never_updated():
    """MF(user_bias=True): the bias cancels in pos - neg; the fused path reports it in the logits and
    leaves it untouched, as dense autograd does (zero gradient)."""
    from revisit_bpr.models import BPR
    from revisit_bpr.models.bpr import MF, set_backend

    U, I, d, B = 50, 40, 16, 32
    data = batches(U, I, B, 2, seed=1)
    outs = {}
    for backend in ("hip", "torch"):
        set_backend(backend)
        try:
            torch.manual_seed(3)
            model = BPR(MF(torch.nn.Embedding(U, d, padding_idx=0), torch.nn.Embedding(I, d, padding_idx=0),
                           item_bias=True, user_bias=True), reg_alphas={"all": 0.01}).cuda()
            with torch.no_grad():
                model.logits_model._user_bias.copy_(torch.linspace(-1, 1, U))
            opt = torch.optim.SGD(model.parameters(), lr=0.1)
            model.train()
            for b in data:
                out = model(b)
                out["loss"].backward()
                opt.step()
                opt.zero_grad()
            outs[backend] = (out["logits_pos"].detach().cpu(), model.logits_model._user_bias.detach().cpu(),
                             model.logits_model._item_emb.weight.detach().cpu())
        finally:
            set_backend("hip")
    assert torch.allclose(outs["hip"][0], outs["torch"][0], atol=1e-5)
    assert torch.equal(outs["hip"][1], torch.linspace(-1, 1, U)) and torch.allclose(outs["hip"][1], outs["torch"][1])
    assert torch.allclose(outs["hip"][2], outs["torch"][2], atol=1e-5)


def _run_loop(model, opt, data, sched=None):
    model.train()
    for b in data:
        out = model(b)
        out["loss"].backward()
        opt.step()
        opt.zero_grad()
        if sched is not None:
            sched.step()


@pytest.mark.parametrize("opt_name", ["adam", "nesterov", "rmsprop"])
@pytest.mark.parametrize("from_torch", [False, True])
def test_checkpoint_resume_continues_the_trajectory(opt_name, from_torch):
    """state_dict() after 4 steps -> fresh model + optimizer -> load_state_dict -> 4 more steps ==
    8 uninterrupted steps: the engine re-binds the LOADED exp_avg / exp_avg_sq /
    momentum_buffer tensors and resumes its step counter from state['step'], so Adam's bias
    corrections and the lazy replay continue at t = 5.  from_torch: the checkpoint is written by
    the dense PyTorch restatement (stock torch.optim state: tensor steps, no step for SGD)."""
    from revisit_bpr.models.bpr import set_backend

    U, I, d, B = 300, 200, 64, 64
    reg = {"user": 0.0016, "item": 0.0001, "neg": 0.00375}
    data = batches(U, I, B, 8, seed=5)
    ref = build(U, I, d, reg, True, seed=11)
    opt = OPTS[opt_name](ref.parameters())
    _run_loop(ref, opt, data)
    want = {k: v.detach().clone() for k, v in ref.state_dict().items()}

    try:
        set_backend("torch" if from_torch else "hip")
        first = build(U, I, d, reg, True, seed=11)
        opt1 = OPTS[opt_name](first.parameters())
        _run_loop(first, opt1, data[:4])
        ck_model = {k: v.detach().clone() for k, v in first.state_dict().items()}
        ck_opt = opt1.state_dict()
    finally:
        set_backend("hip")
    second = build(U, I, d, reg, True, seed=99)  # different init: everything comes from the checkpoint
    second.load_state_dict(ck_model)
    opt2 = OPTS[opt_name](second.parameters())
    opt2.load_state_dict(ck_opt)
    _run_loop(second, opt2, data[4:])
    got = second.state_dict()
    # (torch's SGD keeps no step counter: a loaded momentum_buffer only says "not the first step")
    assert second.engine().step_count == (5 if from_torch and opt_name == "nesterov" else 8)
    atol = 1e-4 if opt_name == "rmsprop" else 2e-5
    for k in want:
        err = (got[k] - want[k]).abs()
        if from_torch and opt_name != "nesterov":
            # dense autograd wrote the checkpoint: Adam / RMSprop elements whose fp32 gradient sum is
            # within rounding of zero differ between the two backends (see close_mostly)
            assert (err > atol).float().mean() <= 1e-3 and err.max() <= 0.1, (k, err.max())
        else:
            assert err.max() <= atol, (k, err.max())
    # the state the optimizer reports is the state the engine wrote
    key = {"adam": "exp_avg", "nesterov": "momentum_buffer", "rmsprop": "square_avg"}[opt_name]
    for p_ref, p_new in zip(ref.parameters(), second.parameters()):
        assert torch.allclose(opt.state[p_ref][key], opt2.state[p_new][key], rtol=0, atol=1e-5)


@pytest.mark.parametrize("opt_name", ["adam", "nesterov"])
def test_lr_schedule_is_honoured_by_the_lazy_replay(opt_name):
    """An LR scheduler changes lr between steps; rows untouched across the change must have their
    missed zero-gradient steps replayed with the lr that was in force at each of them (ADVICE r1:
    flush under the old hyper-parameters before the new ones are bound).  Against dense autograd."""
    from revisit_bpr.models.bpr import set_backend

    U, I, d, B = 300, 200, 32, 16  # small batches: most rows are untouched at every step
    reg = {"user": 0.0016, "item": 0.0001, "neg": 0.00375}
    data = batches(U, I, B, 12, seed=8)
    res = {}
    for backend in ("hip", "torch"):
        set_backend(backend)
        try:
            model = build(U, I, d, reg, False, seed=4)
            opt = OPTS[opt_name](model.parameters())
            sched = torch.optim.lr_scheduler.StepLR(opt, step_size=3, gamma=0.5)
            _run_loop(model, opt, data, sched)
            res[backend] = {k: v.detach().clone() for k, v in model.state_dict().items()}
        finally:
            set_backend("hip")
    for k in res["torch"]:
        assert torch.allclose(res["hip"][k], res["torch"][k], rtol=0, atol=2e-5), \
            (k, (res["hip"][k] - res["torch"][k]).abs().max())


def test_fused_tables_in_separate_param_groups_are_refused():
    U, I, d = 50, 40, 16
    model = build(U, I, d, None, False, seed=1)
    lm = model.logits_model
    opt = torch.optim.SGD([{"params": [lm._user_emb.weight], "lr": 0.1},
                           {"params": [lm._item_emb.weight], "lr": 0.01}])
    b = batches(U, I, 8, 1, seed=2)[0]
    model.train()
    model(b)["loss"].backward()
    with pytest.raises(NotImplementedError):
        opt.step()
    model.engine().discard_grad()


def test_eval_after_forward_without_step_discards_the_pending_gradients():
    U, I, d = 50, 40, 16
    model = build(U, I, d, None, False, seed=1)
    b = batches(U, I, 8, 1, seed=2)[0]
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.train()
    model(b)  # no backward, no step
    model.eval()  # must not leave stale accumulators behind, nor skip the flush silently
    gP, gQ, _ = model.engine().get_grad()
    assert not gP.any() and not gQ.any()
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k])
    model.train()
    model(b)["loss"].backward()
    with pytest.raises(RuntimeError):
        model.eval()  # armed but never stepped
    model.engine().discard_grad()


@pytest.mark.parametrize("kind", ["uniform", "adaptive"])
def test_samplers_with_several_negatives_per_positive(kind):
    """num = batch["item"].size(-1) > 1 (reference: neg_samplers.py:32-37, 76-121): [B, num]
    negatives, never item 0, never a seen item; uniform rows hold DISTINCT items (multinomial draws
    without replacement); and the fused forward / step on the [B, num] batch equals dense autograd."""
    from revisit_bpr.models.bpr import set_backend
    from revisit_bpr.modules import AdaptiveSampler, UniformSampler

    U, I, d, B, num = 200, 120, 32, 64, 3
    reg = {"user": 0.0016, "item": 0.0001, "neg": 0.00375}
    g = torch.Generator().manual_seed(5)
    seen = torch.zeros(U, 12, dtype=torch.long)
    for u in range(1, U):
        k = int(torch.randint(3, 12, (1,), generator=g))
        seen[u, :k] = torch.randperm(I - 1, generator=g)[:k] + 1
    seen = seen.cuda()
    users = torch.randint(1, U, (B,), generator=g).cuda()
    pos = torch.stack([seen[u, torch.randint(0, 3, (num,), generator=g)] for u in users.tolist()])
    model = build(U, I, d, reg, False, seed=2)
    gen = torch.Generator(device="cuda").manual_seed(11)
    batch = {"user": users, "item": pos, "seen_items": seen[users]}
    if kind == "uniform":
        sampler = UniformSampler(I, gen)
    else:
        sampler = AdaptiveSampler(model, I, 0.1, gen, every=10 ** 9)
        sampler.update_stats()
    neg = sampler.sample(batch)
    assert neg.shape == (B, num) and neg.dtype == torch.long
    assert int(neg.min()) >= 1 and int(neg.max()) < I
    assert not bool((neg.unsqueeze(-1) == seen[users].unsqueeze(1)).any())
    if kind == "uniform":
        srt, _ = torch.sort(neg, dim=1)
        assert not bool((srt[:, 1:] == srt[:, :-1]).any())
    batch["neg"] = neg
    res = {}
    for backend in ("hip", "torch"):
        set_backend(backend)
        try:
            m = build(U, I, d, reg, False, seed=2)
            opt = torch.optim.SGD(m.parameters(), lr=0.05)
            m.train()
            out = m(batch)
            assert out["logits"].shape == (B, num)
            out["loss"].backward()
            opt.step()
            m.eval()
            res[backend] = (float(out["loss"].detach()), {k: v.detach().clone() for k, v in m.state_dict().items()})
        finally:
            set_backend("hip")
    assert abs(res["hip"][0] - res["torch"][0]) <= 2e-5 * abs(res["torch"][0])
    for k in res["torch"][1]:
        assert torch.allclose(res["hip"][1][k], res["torch"][1][k], rtol=0, atol=2e-6)


@pytest.mark.parametrize("n,I,tmax,ties", [(300, 977, 40, False), (64, 20108, 700, False), (50, 333, 12, True),
                                           (5, 9000, 5000, False)])
def test_auc_rows_kernel_equals_the_metric_classes(n, I, tmax, ties):
    """`bpr_auc_rows` (r6, csrc/bpr_eval.hip) against the product's sort-based RocAucManySlow — itself pinned to the
    reference's RocAucMany values by tests/golden/metrics.npz — row by row: masked scores (-1e13) count as negatives,
    ties between a positive and a negative are not wins, rows without positives are NaN in both, rows with more
    positives than the kernel's LDS holds come back NaN (evaluate_topk routes them to the sort-based form)."""
    import ctypes

    from revisit_bpr import native
    from revisit_bpr.metrics.auc import RocAucManySlow

    rng = np.random.default_rng(n + I)
    scores = rng.normal(0, 1, (n, I)).astype(np.float32)
    if ties:
        scores = np.round(scores * 2) / 2  # many equal scores
    scores[:, 0] = -1e13
    scores[rng.random((n, I)) < 0.05] = -1e13  # "seen" items
    cnt = rng.integers(0, tmax + 1, n)
    cnt[0] = 0  # a row without positives
    ptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    items = np.concatenate([rng.choice(np.arange(1, I), size=int(c), replace=False) for c in cnt] + [np.zeros(0, np.int64)])
    target = np.zeros((n, I), np.float32)
    for r in range(n):
        target[r, items[ptr[r]:ptr[r + 1]]] = 1.0
    ts, tt = torch.from_numpy(scores).cuda(), torch.from_numpy(target).cuda()
    want = RocAucManySlow().compute(ts, tt).cpu().numpy()
    out = torch.empty(n, device="cuda")
    tp, ti = torch.from_numpy(ptr).cuda(), torch.from_numpy(items.astype(np.int32)).cuda()
    lib = native.load()
    native.check(lib.bpr_auc_rows(ts.data_ptr(), n, I, tp.data_ptr(), ti.data_ptr(), out.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream))
    got = out.cpu().numpy()
    assert np.isnan(got[0]) and np.isnan(want[0])
    big = cnt > 4096
    assert np.isnan(got[big]).all()
    ok = ~big & (cnt > 0)
    assert ok.sum() > 0 and np.abs(got[ok] - want[ok]).max() < 2e-6, np.abs(got[ok] - want[ok]).max()
    assert native.check(lib.bpr_auc_rows(None, 0, I, None, None, None, None)) is None  # n = 0: nothing to do
    with pytest.raises(native.BprError):
        native.check(lib.bpr_auc_rows(None, 3, I, None, None, None, None))
    assert isinstance(ctypes.c_int64(1).value, int)


def test_stream_trainer_launch_split_shares_one_snapshot_per_period(golden_dir):
    """r6 `launch_split`: a refresh period runs as k launches that read the SAME snapshot (a user's triples of a period
    are no longer applied back to back) — k times as many launches of 1 / k the size, the snapshot retaken only before
    the first launch of every period; "auto" = 2 outside the one-rank budget (fast.lag_within_budget), else 1, and 1
    whenever the snapshot is lagged or several ranks share the job."""
    from revisit_bpr import fast
    from revisit_bpr.models import BPR
    from revisit_bpr.models.bpr import MF

    d = np.load(golden_dir / "e2e_data.npz")
    U, I = int(d["num_users"]), int(d["num_items"])
    t = {k: torch.from_numpy(d[k]).cuda() for k in ("users", "items", "indptr", "indices")}

    def trainer(**kw):
        torch.manual_seed(1)
        model = BPR(fuse_forward=True, reg_alphas={"user": 0.001, "item": 0.001, "neg": 0.001},
                    logits_model=MF(torch.nn.Embedding(U, 32, padding_idx=0), torch.nn.Embedding(I, 32, padding_idx=0))).cuda()
        return fast.StreamTrainer(model, t["users"], t["items"], t["indptr"], t["indices"], sampler="adaptive",
                                  adaptive_p=0.05, seed=3, **kw)

    one, two = trainer(lr=0.05, launch_split=1), trainer(lr=0.05, launch_split=2)
    assert two.chunk * 2 == one.chunk and two.rounds in (2 * one.rounds, 2 * one.rounds - 1)
    calls = []
    real = two.engine.adaptive_refresh
    two.engine.adaptive_refresh = lambda: (calls.append(1), real())[1]
    stats = two.train_epoch()
    assert stats["triples"] == t["users"].numel() and len(calls) == -(-two.rounds // 2)  # one snapshot per PERIOD
    # auto: this set's period (10.9 k triples) is inside the budget at lr 0.05, far outside at lr 0.5
    assert trainer(lr=0.05, launch_split="auto").launch_split == 1
    assert trainer(lr=0.5, launch_split="auto").launch_split == 2
    assert trainer(lr=0.5, launch_split="auto", refresh_lag=1.0).launch_split == 1
    with pytest.raises(ValueError):
        trainer(lr=0.05, launch_split=2, refresh_lag=1.0)
