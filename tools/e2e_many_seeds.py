"""Seed-mean parity with MANY of our seeds (GPU box): each path's nDCG@100 / Recall@20 against the
reference's curves (tests/golden/e2e_reference*.json) and against STRICT under the same order
distribution.  The golden protocol trains every reference run on ONE epoch order; paths with their
own device shuffle (STREAM, batched STREAM, "strict-own-order") average over orders, so their
fair yardstick is strict-own-order — exact mini-batch semantics, same order distribution.

  python tools/e2e_many_seeds.py OPT KIND N_SEEDS FIRST_SEED PATH[,PATH...]
    OPT   sgd | adam | rmsprop | nesterov      KIND  uniform | adaptive
    PATH  strict-ref-order | strict-own-order | batched | stream-sync | stream-lag1 |
          stream-lag1-masked | stream-split2-lag1
ref* = the reference over epoch orders (e2e_reference_<opt>_orders.json), ref = its one fixed order.
Results of r03: profiles/r03_strict_adam_uniform_study.txt, profiles/r03_e2e_many_seeds.txt"""
import json
import math
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "revisit-bpr_amd"))
sys.path.insert(0, str(ROOT))
from tests.test_gpu_e2e_parity import SCHEDULES, TORCH_OPT, evaluator, make_model, ref_stats  # noqa: E402
from revisit_bpr.fast import BatchedStreamTrainer, StreamTrainer, StrictTrainer  # noqa: E402

opt_name, kind = sys.argv[1], sys.argv[2]
n_seeds, first = int(sys.argv[3]), int(sys.argv[4])
paths = sys.argv[5].split(",")
g = ROOT / "tests" / "golden"
base = json.loads((g / "e2e_reference.json").read_text())
cfg = base["config"]
if opt_name == "sgd":
    gold, ref = {"lr": cfg["lr"]}, base
else:
    gold = json.loads((g / f"e2e_reference_{opt_name}.json").read_text())
    ref = {"config": cfg, "runs": gold["runs"]}
if (g / f"e2e_reference_{opt_name}_orders.json").exists() and "fixed-order-ref" not in sys.argv:
    # paths with their own shuffle are compared with the reference over epoch orders
    ref_orders = {"config": cfg, "runs": json.loads((g / f"e2e_reference_{opt_name}_orders.json").read_text())["runs"]}
else:
    ref_orders = None
d = np.load(g / "e2e_data.npz")
U, I = int(d["num_users"]), int(d["num_items"])
dev = torch.device("cuda")
users, items = torch.from_numpy(d["users"]).to(dev), torch.from_numpy(d["items"]).to(dev)
indptr, indices = torch.from_numpy(d["indptr"]).to(dev), torch.from_numpy(d["indices"]).to(dev)


def run(path, seed):
    model = make_model(cfg, U, I)
    if opt_name == "sgd":
        opt = torch.optim.SGD(model.parameters(), lr=gold["lr"])
    else:
        opt = TORCH_OPT[opt_name](model.parameters(), gold)
    common = dict(sampler=kind, adaptive_p=cfg["adaptive_p"], batch_size=cfg["B"], seed=seed)
    if path.startswith("strict"):
        tr = StrictTrainer(model, opt, users, items, indptr, indices,
                           order_seed=cfg["order_seed"] if path == "strict-ref-order" else None, **common)
    elif path.startswith("batched"):  # batched | batched-inflight<N>
        cap = int(path[len("batched-inflight"):]) if path.startswith("batched-inflight") else None
        tr = BatchedStreamTrainer(model, opt, users, items, indptr, indices, max_inflight=cap, **common)
    else:
        assert opt_name == "sgd", "STREAM is the plain-SGD path"
        tr = StreamTrainer(model, users, items, indptr, indices, lr=cfg["lr"],
                           **SCHEDULES[path[len("stream-"):]], **common)
    ev = evaluator(model, d)
    curve = [ev()]
    for _ in range(cfg["epochs"]):
        tr.train_epoch()
        curve.append(ev())
    return curve


E = cfg["epochs"]
res = {}
for path in paths:
    curves = [run(path, s) for s in range(first, first + n_seeds)]
    res[path] = curves
    for key in ("ndcg@100", "recall@20"):
        for epoch in (2, 4, E):
            use = ref_orders if (ref_orders is not None and path != "strict-ref-order") else ref
            r = ref_stats(use, kind, key, epoch)
            o = np.array([c[epoch][key] for c in curves])
            se = math.sqrt(r.var(ddof=1) / len(r) + o.var(ddof=1) / len(o))
            line = (f"{path:19s} {opt_name} {kind} {key} epoch {epoch:2d}: ours {o.mean():.4f} (n={len(o)}, sd {o.std(ddof=1):.4f}) "
                    f"ref{'*' if use is ref_orders else ' '} {r.mean():.4f} (n={len(r)})  diff {o.mean() - r.mean():+.4f} z {(o.mean() - r.mean()) / se:+.2f}")
            if "strict-own-order" in res and path != "strict-own-order":
                b = np.array([c[epoch][key] for c in res["strict-own-order"]])
                se2 = math.sqrt(b.var(ddof=1) / len(b) + o.var(ddof=1) / len(o))
                line += f"   vs strict-own-order {o.mean() - b.mean():+.4f} z {(o.mean() - b.mean()) / se2:+.2f}"
            print(line, flush=True)
