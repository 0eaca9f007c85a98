#!/usr/bin/env python
"""Generate the golden vectors that pin the oracle (and, through it, the HIP path).

Runs ONLY in the build container: it imports the reference from /root/reference (read-only, never
copied) together with torch.optim and writes small .npz fixtures next to this file.  The reference
cannot travel to the GPU box; these vectors do.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Contents
  math_<seed>_<variant>.npz  init tables from the reference's MF.reset_parameters, a batch with duplicate
                   users / items / pos-in-one-row-neg-in-another, the reference forward dict, dense
                   autograd grads, and parameters after 1 and 5 steps of every optimizer the
                   reference configs use.
  sampler.npz      _sampling_weights, AdaptiveSampler.update_stats output, and AdaptiveSampler.sample
                   picks with torch.multinomial / geometric_ replaced by injected draws.
  metrics.npz      NDCG / Recall / Precision @k, RocAucOne, RocAucManySlow on fixed logits.
"""
import os
import sys
from pathlib import Path
from unittest import mock

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402
from accelerate.utils import set_seed  # noqa: E402

from revisit_bpr.metrics import (MAP, NDCG, FBeta, Precision, Recall, RocAucMany,  # noqa: E402
                                 RocAucManySlow, RocAucOne)
from revisit_bpr.models import BPR  # noqa: E402
from revisit_bpr.models.bpr import MF  # noqa: E402
from revisit_bpr.modules import AdaptiveSampler  # noqa: E402
from revisit_bpr.modules.neg_samplers import _sampling_weights  # noqa: E402

OUT = Path(__file__).resolve().parent
U, I, D, B, STEPS = 40, 30, 8, 16, 5

OPTIMIZERS = {
    "sgd": lambda p: torch.optim.SGD(p, lr=0.05),
    "sgd_nesterov": lambda p: torch.optim.SGD(p, lr=0.05, momentum=0.9, nesterov=True),
    "sgd_momentum": lambda p: torch.optim.SGD(p, lr=0.05, momentum=0.5),
    "adam_09": lambda p: torch.optim.Adam(p, lr=0.01, betas=(0.9, 0.999)),
    "adam_01": lambda p: torch.optim.Adam(p, lr=0.01, betas=(0.1, 0.999)),
    "adam_00": lambda p: torch.optim.Adam(p, lr=0.01, betas=(0.0, 0.99)),
    "rmsprop": lambda p: torch.optim.RMSprop(p, lr=0.01, alpha=0.9, momentum=0.0),
}

REG_FORMS = {
    "uin": {"user": 0.0016, "item": 0.0001, "neg": 0.00375},
    "all": {"all": 0.00043},
    "item_only": {"item": 0.0025},
    "none": None,
}


def make_model(seed: int, reg, item_bias: bool) -> BPR:
    set_seed(seed)
    return BPR(
        fuse_forward=True,
        logits_model=MF(
            user_emb=torch.nn.Embedding(U, D, padding_idx=0),
            item_emb=torch.nn.Embedding(I, D, padding_idx=0),
            item_bias=item_bias,
            user_bias=False,
        ),
        reg_alphas=reg,
    )


def make_batches(rng: np.random.Generator):
    batches = []
    for _ in range(STEPS):
        users = rng.integers(1, U, size=B)
        pos = rng.integers(1, I, size=B)
        neg = rng.integers(1, I, size=B)
        # duplicates: same user twice, same positive twice, an item positive in one row and negative in another
        users[1] = users[0]
        pos[3] = pos[2]
        neg[5] = pos[4]
        neg[7] = neg[6]
        users[9], pos[9], neg[9] = users[8], pos[8], neg[8]  # fully repeated triple
        for b in range(B):
            while neg[b] == pos[b]:
                neg[b] = rng.integers(1, I)
        batches.append((users.astype(np.int64), pos.astype(np.int64), neg.astype(np.int64)))
    return batches


def batch_dict(users, pos, neg):
    return {
        "user": torch.from_numpy(users),
        "item": torch.from_numpy(pos).unsqueeze(-1),
        "neg": torch.from_numpy(neg).unsqueeze(-1),
    }


def gen_math(seed: int, reg_name: str, item_bias: bool) -> None:
    reg = REG_FORMS[reg_name]
    rng = np.random.default_rng(seed)
    batches = make_batches(rng)
    out = {}
    model = make_model(seed, reg, item_bias)
    feats = model.logits_model.get_features()
    if item_bias:
        # biases start at zero in the reference; give them values so the bias path is exercised
        with torch.no_grad():
            feats["item_bias"].copy_(torch.from_numpy(rng.normal(0, 0.1, I).astype(np.float32)))
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    out["P0"] = feats["user"].detach().numpy().copy()
    out["Q0"] = feats["item"].detach().numpy().copy()
    if item_bias:
        out["b0"] = feats["item_bias"].detach().numpy().copy()
    for s, (u, p, n) in enumerate(batches):
        out[f"users{s}"], out[f"pos{s}"], out[f"neg{s}"] = u, p, n
    # forward + dense grads on batch 0
    model.train()
    o = model(batch_dict(*batches[0]))
    o["loss"].backward()
    for k in ("logits_pos", "logits_neg", "logits", "bpr_loss", "l2_reg", "loss"):
        out[f"fwd_{k}"] = o[k].detach().numpy().copy()
    out["gP"] = feats["user"].grad.numpy().copy()
    out["gQ"] = feats["item"].grad.numpy().copy()
    if item_bias:
        out["gb"] = feats["item_bias"].grad.numpy().copy()
    # optimizers
    for name, ctor in OPTIMIZERS.items():
        model = make_model(seed, reg, item_bias)
        model.load_state_dict(init)
        model.train()
        opt = ctor(model.parameters())
        feats = model.logits_model.get_features()
        for s in range(STEPS):
            o = model(batch_dict(*batches[s]))
            o["loss"].backward()
            opt.step()
            opt.zero_grad()
            if s in (0, STEPS - 1):
                out[f"{name}_P{s + 1}"] = feats["user"].detach().numpy().copy()
                out[f"{name}_Q{s + 1}"] = feats["item"].detach().numpy().copy()
                if item_bias:
                    out[f"{name}_b{s + 1}"] = feats["item_bias"].detach().numpy().copy()
            out[f"{name}_loss{s + 1}"] = o["loss"].detach().numpy().copy()
    tag = f"math_{seed}_{reg_name}_{'bias' if item_bias else 'nobias'}"
    np.savez_compressed(OUT / f"{tag}.npz", **out)
    print("wrote", tag, len(out), "arrays")


def gen_sampler() -> None:
    seed = 13
    rng = np.random.default_rng(seed)
    Us, Is, Ds, Bs, S = 24, 50, 6, 12, 9
    set_seed(seed)
    model = BPR(
        fuse_forward=True,
        logits_model=MF(torch.nn.Embedding(Us, Ds, padding_idx=0),
                        torch.nn.Embedding(Is, Ds, padding_idx=0)),
    )
    feats = model.logits_model.get_features()
    users = rng.integers(1, Us, size=Bs).astype(np.int64)
    # padded seen matrix (0 = pad), ragged lengths incl. an empty row and a full-width row
    seen = np.zeros((Bs, S), np.int64)
    for b in range(Bs):
        n = [0, S, 3, 5][b % 4] if b < 4 else int(rng.integers(1, S + 1))
        seen[b, :n] = rng.choice(np.arange(1, Is), size=n, replace=False)
    # same user → same seen row
    for b in range(Bs):
        for c in range(b):
            if users[c] == users[b]:
                seen[b] = seen[c]
    out = {"P": feats["user"].detach().numpy().copy(), "Q": feats["item"].detach().numpy().copy(),
           "users": users, "seen": seen}
    base = torch.ones(Is)
    out["weights"] = _sampling_weights(base, torch.from_numpy(seen)).numpy()
    sampler = AdaptiveSampler(model, num_items=Is, sampling_prob=0.1,
                              neg_gen=torch.Generator().manual_seed(seed), every=10**9)
    sampler.update_stats()
    out["factor_to_items"] = sampler._factor_to_items.numpy().copy()
    out["factor_std"] = sampler._factor_std.numpy().copy()
    # injected draws: every factor × a spread of geometric draws r (1-based, some beyond #unseen)
    picks, facs, rs = [], [], []
    for f in range(Ds):
        for r in (1, 2, 3, 7, 20, 39, 40, 45, 200):
            factor = torch.full((Bs, 1), f, dtype=torch.long)
            r_t = torch.full((Bs, 1), float(r))

            def fake_multinomial(*a, **k):
                return factor.clone()

            def fake_geometric(self, *a, **k):
                return self.copy_(r_t.to(self.dtype))

            with mock.patch.object(torch, "multinomial", fake_multinomial), \
                    mock.patch.object(torch.Tensor, "geometric_", fake_geometric):
                neg = sampler.sample({"user": torch.from_numpy(users),
                                      "item": torch.zeros(Bs, 1, dtype=torch.long),
                                      "seen_items": torch.from_numpy(seen)})
            picks.append(neg.numpy().reshape(-1).copy())
            facs.append(f)
            rs.append(r)
    out["inj_factor"] = np.asarray(facs, np.int64)
    out["inj_r"] = np.asarray(rs, np.int64)
    out["inj_picks"] = np.stack(picks)  # [cases, B]
    np.savez_compressed(OUT / "sampler.npz", **out)
    print("wrote sampler", out["inj_picks"].shape)


def gen_metrics() -> None:
    rng = np.random.default_rng(7)
    out = {}
    for name, (nb, ni) in {"wide": (9, 300), "narrow": (5, 12)}.items():
        logits = rng.normal(size=(nb, ni)).astype(np.float32)
        target = (rng.random((nb, ni)) < 0.08).astype(np.float32)
        target[0] = 0.0  # user without positives → nan_to_num path
        target[1, :] = 0.0
        target[1, 3] = 1.0
        logits[2, :20] = -1e13  # masked (seen) items
        out[f"{name}_logits"], out[f"{name}_target"] = logits, target
        lt, tt = torch.from_numpy(logits), torch.from_numpy(target)
        for k in (5, 10, 20, 50, 100):
            out[f"{name}_ndcg@{k}"] = NDCG(topk=k).compute(lt, tt).numpy()
            out[f"{name}_recall@{k}"] = Recall(topk=k).compute(lt, tt).numpy()
            out[f"{name}_precision@{k}"] = Precision(topk=k).compute(lt, tt).numpy()
        out[f"{name}_auc_many"] = RocAucManySlow().compute(lt, tt).numpy()
        out[f"{name}_auc_one"] = RocAucOne().compute(lt, tt).numpy()
        m = NDCG(topk=10)
        m(lt, tt)
        m(lt[:3], tt[:3])
        out[f"{name}_ndcg@10_stream"] = m.get_metric().numpy()
        # r4: the remaining metric classes (map.py, fbeta.py, auc.py RocAucMany, NDCG's linear gain)
        # and the mask argument of the AUC family; the mask comes from its own stream so that the
        # arrays above stay what they were
        mrng = np.random.default_rng(11 + nb)
        mask = (mrng.random((nb, ni)) < 0.85).astype(np.float32)
        mask[:, 0] = 1.0
        out[f"{name}_mask"] = mask
        mt = torch.from_numpy(mask)
        for k in (5, 10, 20, 50, 100):
            out[f"{name}_map@{k}"] = MAP(topk=k).compute(lt, tt).numpy()
            out[f"{name}_map_raw@{k}"] = MAP(topk=k, normalized=False).compute(lt, tt).numpy()
            out[f"{name}_f1@{k}"] = FBeta(topk=k).compute(lt, tt).numpy()
            out[f"{name}_f0.5@{k}"] = FBeta(topk=k, beta=0.5).compute(lt, tt).numpy()
            out[f"{name}_ndcg_linear@{k}"] = NDCG(topk=k, gain_function="linear").compute(lt, tt).numpy()
        out[f"{name}_auc_many_dense"] = RocAucMany().compute(lt, tt).numpy()
        out[f"{name}_auc_many_dense_masked"] = RocAucMany().compute(lt, tt, mt).numpy()
        out[f"{name}_auc_many_masked"] = RocAucManySlow().compute(lt, tt, mt).numpy()
        out[f"{name}_auc_one_masked"] = RocAucOne().compute(lt, tt, mt).numpy()
        for cls, key in ((MAP, "map"), (FBeta, "f1"), (Recall, "recall"), (Precision, "precision")):
            m = cls(topk=10)
            m(lt, tt)
            m(lt[:3], tt[:3])
            out[f"{name}_{key}@10_stream"] = m.get_metric().numpy()
        # the eval loop's view (example.py:195-230): item 0 masked like a seen item, never a target
        l0, t0 = lt.clone(), tt.clone()
        l0[:, 0] = -1e13
        t0[:, 0] = 0.0
        for k in (5, 10, 20, 50, 100):
            out[f"{name}_pad0_ndcg@{k}"] = NDCG(topk=k).compute(l0, t0).numpy()
            out[f"{name}_pad0_recall@{k}"] = Recall(topk=k).compute(l0, t0).numpy()
            out[f"{name}_pad0_precision@{k}"] = Precision(topk=k).compute(l0, t0).numpy()
        out[f"{name}_pad0_auc_many"] = RocAucManySlow().compute(l0, t0).numpy()
        m = RocAucManySlow()
        m(lt[1:], tt[1:], mt[1:])  # (row 0 has no positives: 0/0)
        m(lt[1:4], tt[1:4])
        out[f"{name}_auc_many_stream"] = m.get_metric().numpy()
    np.savez_compressed(OUT / "metrics.npz", **out)
    print("wrote metrics")


def gen_knn() -> None:
    """The reference's two item-to-item scorers (models/bpr/model.py:156-255) on a small batch with
    candidates that coincide with seen items -> tests/golden/knn.npz (weights, inputs, logits)."""
    from revisit_bpr.models.bpr import FreeItemKNN, ItemKNN

    rng = np.random.default_rng(3)
    n_items, hidden, B, C, S = 40, 8, 5, 6, 7
    item = rng.integers(1, n_items, (B, C))
    seen = rng.integers(0, n_items, (B, S))
    seen[:, :2] = item[:, :2]  # seen items among the candidates: masked
    out = {"item": item, "seen": seen}
    for name, mod in (("knn", ItemKNN(n_items, hidden, bias=True)), ("free", FreeItemKNN(n_items, bias=True))):
        with torch.no_grad():
            mod._weights.copy_(torch.from_numpy(rng.normal(0, 1, tuple(mod._weights.shape)).astype(np.float32)))
            mod._bias.copy_(torch.from_numpy(rng.normal(0, 1, n_items).astype(np.float32)))
            logits = mod(None, torch.from_numpy(item), {"seen_items": torch.from_numpy(seen)})
        out[f"{name}_w"] = mod._weights.detach().numpy()
        out[f"{name}_b"] = mod._bias.detach().numpy()
        out[f"{name}_logits"] = logits.numpy()
    np.savez_compressed(OUT / "knn.npz", **out)
    print("wrote knn")


if __name__ == "__main__":
    torch.set_num_threads(1)
    if sys.argv[1:] == ["metrics"]:  # only metrics.npz
        gen_metrics()
        sys.exit(0)
    if sys.argv[1:] == ["knn"]:  # only knn.npz
        gen_knn()
        sys.exit(0)
    for seed in (13, 42069):
        gen_math(seed, "uin", False)
    gen_math(13, "uin", True)
    gen_math(13, "all", False)
    gen_math(13, "item_only", True)
    gen_math(42069, "none", False)
    gen_sampler()
    gen_metrics()
    gen_knn()
