#!/usr/bin/env python
"""N-rank STREAM training on the e2e parity set (tests/golden/e2e_data.npz) — user shards + ItemSync —
to compare nDCG@100 / Recall@20 with the reference's single-process curves.  Launch with
torch.distributed.run; rank 0 prints one JSON line per seed.  BPR_DIST_BACKEND=gloo lets several
ranks share one GPU (functional/parity check of the protocol; RCCL needs a device per rank)."""
import json
import os
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "revisit-bpr_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from revisit_bpr.distributed import ItemSync, balanced_user_shards, owner_of  # noqa: E402
from revisit_bpr.evaluation import evaluate_topk  # noqa: E402
from revisit_bpr.fast import BatchedStreamTrainer, StreamTrainer, StrictTrainer  # noqa: E402
from revisit_bpr.models import BPR  # noqa: E402
from revisit_bpr.models.bpr import MF  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
    seeds = [int(s) for s in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "2", "3"])]
    # "stream" (SGD, fused kernel) | "stream-lag" (the same with the overlapped snapshot schedule) |
    # "stream-shard" (the snapshot sort shared by the ranks + all-gather) |
    # "adam" / "sgd" (STRICT mini-batches) | "batched-adam" (single-launch Adam: BASELINE configs[4])
    mode = sys.argv[3] if len(sys.argv) > 3 else "stream"
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        backend = os.environ.get("BPR_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    ref = json.loads((ROOT / "tests/golden/e2e_reference.json").read_text())
    cfg = ref["config"]
    if len(sys.argv) > 4 and sys.argv[4] == "full":  # ML-20M-shaped latent-structure set, d=128
        from revisit_bpr.datasets import synthetic

        data = synthetic.generate_latent(136677, 20108, 9_700_000, factors=16, strength=1.2,
                                         median_per_user=37, min_per_user=5, seed=13,
                                         eval_users=10_000, item_skew=1.2, item_shift=60.0, cache_dir=tempfile.gettempdir())
        d = {k: getattr(data, k) for k in ("users", "items", "indptr", "indices", "eval_users",
                                           "eval_indptr", "eval_items")}
        d["num_users"], d["num_items"] = data.num_users, data.num_items
        cfg = dict(cfg, d=128, B=256, epochs=int(os.environ.get("BPR_EPOCHS", "4")), lr=float(os.environ.get('BPR_LR', '0.05')),
                   adaptive_p=0.01, reg={"user": 0.0016, "item": 0.0001, "neg": 0.00375})
    else:
        d = np.load(ROOT / "tests/golden/e2e_data.npz")
    U, I = int(d["num_users"]), int(d["num_items"])
    t = {k: torch.from_numpy(d[k]).to(dev) for k in ("users", "items", "indptr", "indices", "eval_users",
                                                     "eval_indptr", "eval_items")}
    bounds = balanced_user_shards(d["indptr"], world)
    mine = torch.from_numpy(owner_of(d["users"], bounds) == rank).to(dev)
    for seed in seeds:
        torch.manual_seed(cfg["init_seed"])
        model = BPR(fuse_forward=True, reg_alphas=cfg["reg"],
                    logits_model=MF(torch.nn.Embedding(U, cfg["d"], padding_idx=0),
                                    torch.nn.Embedding(I, cfg["d"], padding_idx=0))).to(dev)
        f = model.logits_model.get_features()
        # BPR_HOT_ROWS / BPR_HOT_SPLIT: the two-tier reconciliation (hot block after every launch)
        hot_rows, hot_split = int(os.environ.get("BPR_HOT_ROWS", "0")), int(os.environ.get("BPR_HOT_SPLIT", "1"))
        sync = ItemSync([f["item"].data], engine=model.engine(), hot_rows=hot_rows,
                        local_items=t["items"][mine]) if world > 1 else None
        if mode == "batched-adam":
            opt = torch.optim.Adam(model.parameters(), lr=float(os.environ.get("BPR_ADAM_LR", "0.002")))
            tr = BatchedStreamTrainer(model, opt, t["users"][mine].contiguous(),
                                      t["items"][mine].contiguous(), t["indptr"], t["indices"],
                                      sampler=kind, adaptive_p=cfg["adaptive_p"], batch_size=cfg["B"],
                                      seed=seed, rank=rank, item_sync=sync)
        elif mode in ("adam", "sgd"):
            opt = (torch.optim.Adam(model.parameters(), lr=float(os.environ.get("BPR_ADAM_LR", "0.002")))
                   if mode == "adam" else torch.optim.SGD(model.parameters(), lr=cfg["lr"]))
            tr = StrictTrainer(model, opt, t["users"][mine].contiguous(), t["items"][mine].contiguous(),
                               t["indptr"], t["indices"], sampler=kind, adaptive_p=cfg["adaptive_p"],
                               batch_size=cfg["B"], seed=seed, rank=rank, item_sync=sync)
        else:
            tr = StreamTrainer(model, t["users"][mine].contiguous(), t["items"][mine].contiguous(),
                               t["indptr"], t["indices"], lr=cfg["lr"], sampler=kind,
                               adaptive_p=cfg["adaptive_p"], batch_size=cfg["B"], seed=seed, rank=rank,
                               item_sync=sync,
                               # BPR_CADENCE=rank: a full refresh period per rank and chunk (the
                               # default divides the period by the number of ranks)
                               cadence=os.environ.get("BPR_CADENCE", "job"), hot_split=hot_split,
                               **({"refresh_lag": 1.0, "refresh_cus": 64} if mode == "stream-lag" else {}),
                               **({"shard_refresh": True} if mode == "stream-shard" else {}))
        curve = []
        for _ in range(cfg["epochs"]):
            tr.train_epoch()
            if world > 1:
                for r in range(world):
                    lo, hi = int(bounds[r]), int(bounds[r + 1])
                    if hi > lo:
                        dist.broadcast(f["user"].data[lo:hi], src=r)
            if mode in ("adam", "sgd", "batched-adam"):
                model.sync()  # bring lazily-updated rows to "now" before reading the tables
            if rank == 0:
                m = evaluate_topk(f["user"].data, f["item"].data, None, t["eval_users"], t["eval_indptr"],
                                  t["eval_items"], t["indptr"], t["indices"], ks=(20, 100))
                curve.append((m["ndcg@100"], m["recall@20"]))
        if sync is not None:
            sync.close()
        if rank == 0:
            print(json.dumps({"kind": kind, "seed": seed, "world": world, "mode": mode,
                              "ndcg@100": [c[0] for c in curve], "recall@20": [c[1] for c in curve]}),
                  flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
