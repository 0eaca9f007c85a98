#!/usr/bin/env python
"""The metric's SHAPE pinned to the reference itself: the REFERENCE's training loop
(/root/reference/example.py:157-192 — `neg_sampler.sample` -> `model(batch)` -> `backward` ->
`optimizer.step`, AdaptiveSampler of revisit_bpr/modules/neg_samplers.py:74-132) imported in place
(build container only) and run for a PREFIX of the first epoch on the ML-20M-shaped synthetic set the
full-scale gates use (136,677 x 20,108, ~9.6 M training triples, d = 128, SGD lr 0.05, B = 256,
adaptive p = 1/100, the L2 of configs/RQ2/neg-sampling/ada-sampling-ml-20m.yaml.j2), nDCG@100 /
Recall@20 on the 10,000 held-out users at the checkpoints — whole refresh periods of the sampler
(778 batches = 199,168 triples each), so that the STREAM path's launches end exactly there.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_fullscale.py run <sampler seed> [threads]
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_fullscale.py merge

`run` writes tests/golden/e2e_ml20m_reference_prefix_<seed>.json (after every checkpoint, so a run
can be read while it goes on); `merge` folds the per-seed files into e2e_ml20m_reference_prefix.json.
The reference does ~1.5 k triples/s here: half an epoch is about an hour per seed.  The dataset is
not stored (our seeded generator): the fixture carries its checksum.
"""
import hashlib
import json
import math
import os
import sys
import tempfile
import time
from pathlib import Path

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, "/root/reference")
sys.path.insert(1, str(ROOT / "revisit-bpr_amd" / "revisit_bpr" / "datasets"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
from accelerate.utils import set_seed  # noqa: E402

import synthetic  # noqa: E402  (our generator, by path: `revisit_bpr` below is the REFERENCE package)
from revisit_bpr.metrics import NDCG, Recall  # noqa: E402
from revisit_bpr.models import BPR  # noqa: E402
from revisit_bpr.models.bpr import MF  # noqa: E402
from revisit_bpr.modules import AdaptiveSampler  # noqa: E402

OUT = Path(__file__).resolve().parent
GEN = dict(users=136677, items=20108, actions=9_700_000, factors=16, strength=1.2, median_per_user=37,
           min_per_user=5, seed=13, eval_users=10_000, item_skew=1.2, item_shift=60.0)
D, B, LR, P_GEO = 128, 256, 0.05, 0.01
REG = {"user": 0.0016, "item": 0.0001, "neg": 0.00375}
INIT_SEED, ORDER_SEED = 13, 13
# checkpoints in refresh periods (I ln I / B = 778 batches): ~0.26 and ~0.51 epoch; E2E_PERIODS=12,24,36,47
# runs on to the end of the first epoch (47 whole periods = 0.996 epoch)
CHECKPOINT_PERIODS = tuple(int(x) for x in os.environ.get("E2E_PERIODS", "12,24").split(","))


def dataset():
    return synthetic.generate_latent(cache_dir=tempfile.gettempdir(), **GEN)


def data_checksum(data):
    h = hashlib.sha256()
    for a in (data.users, data.items, data.indptr, data.indices, data.eval_users, data.eval_indptr,
              data.eval_items):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def padded_seen(data, users):
    """[len(users), max seen among them] int64, zero-padded: what Collator(pad=["seen_items"]) hands
    the sampler (example.py:314)"""
    lo, hi = data.indptr[users], data.indptr[users + 1]
    S = int((hi - lo).max())
    out = np.zeros((len(users), max(S, 1)), np.int64)
    for r in range(len(users)):
        out[r, :hi[r] - lo[r]] = data.indices[lo[r]:hi[r]]
    return torch.from_numpy(out)


@torch.no_grad()
def evaluate(model, data):
    model.eval()
    nd, rc = NDCG(topk=100), Recall(topk=20)
    items = torch.arange(data.num_items)
    for lo in range(0, len(data.eval_users), 512):
        eu = data.eval_users[lo:lo + 512].astype(np.int64)
        tgt = torch.zeros(len(eu), data.num_items)
        for r in range(len(eu)):
            tgt[r, data.eval_items[data.eval_indptr[lo + r]:data.eval_indptr[lo + r + 1]]] = 1.0
        logits = model({"user": torch.from_numpy(eu), "item": items.expand(len(eu), -1)})["logits"]
        logits.scatter_(-1, padded_seen(data, eu), -1e13)  # example.py:215-217
        logits[:, 0] = -1e13
        nd(logits, tgt)
        rc(logits, tgt)
    model.train()
    return float(nd.get_metric()), float(rc.get_metric())


def run(seed, threads):
    torch.set_num_threads(threads)
    data = dataset()
    every = int(data.num_items * math.log(data.num_items) / B)  # example.py:302
    set_seed(INIT_SEED)
    model = BPR(fuse_forward=True, reg_alphas=REG,
                logits_model=MF(torch.nn.Embedding(data.num_users, D, padding_idx=0),
                                torch.nn.Embedding(data.num_items, D, padding_idx=0)))
    opt = torch.optim.SGD(model.parameters(), lr=LR)
    sampler = AdaptiveSampler(model, data.num_items, P_GEO, torch.Generator().manual_seed(seed), every=every)
    sampler.update_stats()
    users_t = torch.from_numpy(data.users.astype(np.int64))
    items_t = torch.from_numpy(data.items.astype(np.int64))
    perm = np.random.default_rng(ORDER_SEED).permutation(data.nnz)  # DataLoader(shuffle=True) stand-in
    out = {"seed": seed, "checkpoints": {}, "threads": threads}
    path = OUT / f"e2e_ml20m_reference_prefix_{seed}{'' if len(CHECKPOINT_PERIODS) <= 2 else '_long'}.json"
    t0 = time.time()
    nd0, rc0 = evaluate(model, data)
    out["checkpoints"]["0"] = {"batches": 0, "triples": 0, "ndcg@100": nd0, "recall@20": rc0}
    model.train()
    done = 0
    for periods in CHECKPOINT_PERIODS:
        stop = periods * every
        for b in range(done, stop):
            idx = perm[b * B:(b + 1) * B]
            u = data.users[idx].astype(np.int64)
            batch = {"user": users_t[idx], "item": items_t[idx].unsqueeze(-1), "seen_items": padded_seen(data, u)}
            batch["neg"] = sampler.sample(batch)
            loss = model(batch)["loss"]
            loss.backward()
            opt.step()
            opt.zero_grad()
            if b % 200 == 0:
                print(f"seed {seed}: batch {b}/{CHECKPOINT_PERIODS[-1] * every} {time.time() - t0:.0f}s "
                      f"loss/triple {loss.item() / B:.4f}", flush=True)
        done = stop
        nd, rc = evaluate(model, data)
        out["checkpoints"][str(periods)] = {"batches": stop, "triples": stop * B, "ndcg@100": nd, "recall@20": rc,
                                            "seconds": time.time() - t0}
        path.write_text(json.dumps(out, indent=1))
        print(f"seed {seed}: {periods} periods ({stop * B} triples) nDCG@100 {nd:.4f} Recall@20 {rc:.4f} "
              f"{time.time() - t0:.0f}s", flush=True)


def merge():
    data = dataset()
    every = int(data.num_items * math.log(data.num_items) / B)
    res = {"config": {"generator": "synthetic.generate_latent(" + ", ".join(f"{k}={v}" for k, v in GEN.items()) + ")",
                      "data_sha256": data_checksum(data), "train_triples": int(data.nnz), "d": D, "B": B, "lr": LR,
                      "adaptive_p": P_GEO, "reg": REG, "init_seed": INIT_SEED, "order_seed": ORDER_SEED,
                      "order": "np.random.default_rng(order_seed).permutation(train_triples), first epoch prefix",
                      "refresh_every_batches": every, "checkpoint_periods": list(CHECKPOINT_PERIODS),
                      "eval_users": int(len(data.eval_users)),
                      "loop": "/root/reference/example.py:157-192 with revisit_bpr.modules.AdaptiveSampler"},
           "runs": {}}
    main_file = OUT / "e2e_ml20m_reference_prefix.json"
    if main_file.exists():
        res["runs"] = json.loads(main_file.read_text())["runs"]
    for f in sorted(OUT.glob("e2e_ml20m_reference_prefix_*.json")):
        j = json.loads(f.read_text())
        old = res["runs"].get(str(j["seed"]), {})
        if len(j["checkpoints"]) >= len(old):  # a longer run of the same seed replaces a shorter one
            for k, v in old.items():           # ... and must reproduce it (torch CPU, same thread count)
                assert k not in j["checkpoints"] or abs(j["checkpoints"][k]["ndcg@100"] - v["ndcg@100"]) < 1e-6, (f, k)
            res["runs"][str(j["seed"])] = j["checkpoints"]
        f.unlink()
    res["config"]["checkpoint_periods"] = sorted({int(k) for r in res["runs"].values() for k in r})
    main_file.write_text(json.dumps(res, indent=1))
    print(main_file.name, sorted(res["runs"]))


if __name__ == "__main__":
    if sys.argv[1:2] == ["merge"]:
        merge()
    else:
        run(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 2)
