"""Config loading (jinja2 → YAML → instantiate) and the dataset / collator classes named by the
reference's configs — host logic, no GPU."""
from pathlib import Path

import numpy as np
import torch

from experiments.bpr.dataset import (AllItemsCollator, InMemory, ManyPosCollator, OnePosCollator,
                                     SparseSamplingInMemoryWithCollator)
from experiments.config import instantiate, parse_extra_vars, render
from revisit_bpr.datasets import interactions, synthetic

CONFIG = Path(__file__).parent / "configs" / "bpr_small.yaml.j2"


def test_extra_vars_and_render(tmp_path):
    v = parse_extra_vars("dataset=/d/x;num_users=10; num_items=7;embedding_dim=8;train_batch_size=4")
    assert v["dataset"] == "/d/x" and v["num_items"] == "7"
    cfg = render(CONFIG, v)
    assert cfg["num_users"] == 11 and cfg["num_items"] == 8 and cfg["epochs"] == 3
    assert cfg["model"]["logits_model"]["user_emb"]["num_embeddings"] == 11  # YAML anchors survive
    assert cfg["experiment"]["_target_"] == "experiments.bpr.Experiment"


def test_instantiate_targets_partials_and_nesting():
    opt = instantiate({"_partial_": True, "_target_": "torch.optim.Adam", "lr": 0.01, "betas": [0.1, 0.999]})
    lin = instantiate({"_target_": "torch.nn.Linear", "in_features": 3, "out_features": 2})
    o = opt(lin.parameters())
    assert isinstance(o, torch.optim.Adam) and tuple(o.param_groups[0]["betas"]) == (0.1, 0.999)
    nested = instantiate({"a": {"_target_": "torch.nn.Embedding", "num_embeddings": 5, "embedding_dim": 2,
                                "_convert_": "all"}, "b": [1, {"_target_": "builtins.dict", "x": 1}]})
    assert isinstance(nested["a"], torch.nn.Embedding) and nested["b"][1] == {"x": 1}
    cfg = render(CONFIG, parse_extra_vars("dataset=/x;num_users=20;num_items=9;embedding_dim=4;train_batch_size=2"))
    model = instantiate(cfg["model"])
    assert type(model).__name__ == "Model" and model.logits_model._item_bias is not None
    metrics = instantiate(cfg["experiment"]["metrics"])
    assert sorted(metrics) == ["auc", "ndcg@100", "precision@10", "recall@20"]


def test_datasets_and_collators(tmp_path):
    data = synthetic.generate_latent(60, 40, 900, seed=2)
    interactions.write_dataset(data, tmp_path)
    ds = SparseSamplingInMemoryWithCollator(tmp_path / interactions.TRAIN, tmp_path / interactions.SEEN,
                                            data.num_users, data.num_items)
    assert len(ds) == data.nnz and ds[5] == 5
    b = ds.collate_fn([0, 3, 7])
    assert set(b) == {"user", "item", "seen_items"} and b["seen_items"].shape[0] == 3
    u0 = int(b["user"][0])
    row = b["seen_items"][0]
    assert sorted(row[row != 0].tolist()) == data.indices[data.indptr[u0]:data.indptr[u0 + 1]].tolist()
    indptr, indices = ds.seen_csr()
    assert np.array_equal(indptr.numpy(), data.indptr) and np.array_equal(indices.numpy(), data.indices)
    ev = InMemory(tmp_path / interactions.TEST, tmp_path / interactions.SEEN)
    batch = AllItemsCollator(data.num_items)([ev[0], ev[1]])
    assert batch["item"].shape == (2, data.num_items) and batch["target"].sum() == len(ev[0]["item"]) + len(ev[1]["item"])
    assert batch["seen_items"].shape[0] == 2
    many = ManyPosCollator(data.num_items)([ev[0], ev[1]])
    n0 = len(ev[0]["item"])
    assert many["target"][0, :n0].sum() == n0 and many["target"][0, n0:].sum() == 0
    assert set(many) == {"user", "item", "seen_items", "target", "mask"}
    # no seen item among the negatives of a ManyPos row
    negs = many["item"][0, n0:][many["mask"][0, n0:] > 0]
    assert not set(negs.tolist()) & set(ev[0]["seen_items"])
    one = OnePosCollator(10)([{"user": 3, "item": 1, "seen_items": [4, 7, 2]}])
    assert one["item"][0, 0] == 7 and one["target"][0, 0] == 1 and one["item"].shape[1] == 1 + 10 - 1 - 3


REFERENCE_CONFIGS = Path("/root/reference/configs")


import pytest  # noqa: E402


@pytest.mark.skipif(not REFERENCE_CONFIGS.exists(), reason="reference tree only exists in the build container")
def test_reference_bpr_configs_load_unchanged():
    """Every reference config that targets revisit_bpr.models.bpr.Model renders, parses and
    instantiates (experiment object, model, optimizer partial, metrics) with this repository's
    classes — read in place from /root/reference, nothing is copied."""
    common = ("dataset=/data/x;num_users=50;num_items=40;embedding_dim=16;train_batch_size=8;"
              "eval_batch_size=8;epochs=1;item_bias=false;seed=13;exp_name=t;neg_sampling_alpha=0;"
              "lr=0.01;max_iters=;device=cpu")
    loaded = 0
    skipped = []
    for path in sorted(REFERENCE_CONFIGS.rglob("*.yaml.j2")):
        text = path.read_text()
        if "revisit_bpr.models.bpr.Model" not in text:
            continue
        try:
            cfg = render(path, parse_extra_vars(common))
        except Exception as exc:  # a template may need variables outside the common set
            skipped.append((path.name, type(exc).__name__))
            continue
        assert cfg["experiment"]["_target_"] == "experiments.bpr.Experiment"
        model = instantiate(cfg["model"])
        assert type(model).__name__ == "Model" and type(model.logits_model).__name__ == "MF"
        opt = instantiate(cfg["optimizer"])(model.parameters())
        assert isinstance(opt, torch.optim.Optimizer)
        exp_cfg = dict(cfg.pop("experiment"))
        try:
            exp = instantiate(exp_cfg, exp_config=lambda c=cfg: c, dir=None, seed=13, debug=False)
        except NotImplementedError:
            skipped.append((path.name, "neg_sampling_alpha"))
            continue
        assert hasattr(exp, "run") and all(hasattr(m, "compute") for m in exp._metrics.values())
        loaded += 1
    assert loaded >= 15, (loaded, skipped)


def test_generate_latent_cache_round_trip(tmp_path):
    """`cache_dir` returns the arrays generate_latent computes (same values and dtypes), from the
    file on the second call; other arguments get another file."""
    args = dict(factors=8, seed=3, eval_users=100)
    a = synthetic.generate_latent(3000, 500, 60000, **args)
    b = synthetic.generate_latent(3000, 500, 60000, cache_dir=tmp_path, **args)
    files = sorted(p.name for p in tmp_path.iterdir())
    assert len(files) == 1 and files[0].startswith("bpr_latent_") and files[0].endswith(".npz")
    c = synthetic.generate_latent(3000, 500, 60000, cache_dir=tmp_path, **args)
    for k in ("users", "items", "indptr", "indices", "eval_users", "eval_indptr", "eval_items"):
        for other in (b, c):
            assert np.array_equal(getattr(a, k), getattr(other, k)) and getattr(a, k).dtype == getattr(other, k).dtype, k
    assert (a.num_users, a.num_items) == (c.num_users, c.num_items) and isinstance(c.num_users, int)
    synthetic.generate_latent(3000, 500, 60000, cache_dir=tmp_path, **dict(args, seed=4))
    assert len(list(tmp_path.iterdir())) == 2


def test_item_knn_scorers_match_the_reference(golden_dir):
    """revisit_bpr.models.bpr exports the reference's names (models/bpr/__init__.py:1-8); the two
    item-to-item scorers are plain PyTorch modules held to logits of the reference's own
    (tests/golden/knn.npz from make_golden.py knn), candidates that are seen items included."""
    import numpy as np
    import torch

    from revisit_bpr.models.bpr import BaseLogitModel, FreeItemKNN, ItemKNN, Loss, MF, Model  # noqa: F401

    g = np.load(golden_dir / "knn.npz")
    item, seen = torch.from_numpy(g["item"]), torch.from_numpy(g["seen"])
    for name, mod in (("knn", ItemKNN(40, 8, bias=True)), ("free", FreeItemKNN(40, bias=True))):
        assert float(mod._weights[0].abs().sum()) == 0.0  # the pad row
        with torch.no_grad():
            mod._weights.copy_(torch.from_numpy(g[f"{name}_w"]))
            mod._bias.copy_(torch.from_numpy(g[f"{name}_b"]))
            out = mod(None, item, {"seen_items": seen})
        assert np.allclose(out.numpy(), g[f"{name}_logits"], rtol=1e-5, atol=1e-5)
        assert set(mod.get_features()) == {"item", "bias"}
