#!/usr/bin/env python
"""Train BPR-MF on a JSONL dataset directory with the MI355X engine — same command line as the
reference's example.py (`python example.py DATASET_PATH [--num-users ...]`), same files, same
hyper-parameters, same metrics; the epoch loop runs on the device (revisit_bpr.fast.StreamTrainer)
or, with --mode strict, through the reference's batch loop (sampler → model → backward → step).

    python example.py /data/ml-20m --embedding-dim 1024 --epochs 72
    python example.py --synthetic ml-20m --embedding-dim 128 --epochs 2      # no files needed
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 example.py ...
        # one rank per GPU: users sharded by interaction count, item table replicated and
        # reconciled by an asynchronous all-reduce of item deltas (revisit_bpr/distributed.py)
"""
from __future__ import annotations

import logging
import math
import time
from pathlib import Path

import click
import torch

from revisit_bpr.datasets import interactions, synthetic
from revisit_bpr.evaluation import evaluate
from revisit_bpr.fast import StreamTrainer
from revisit_bpr.metrics import NDCG, Precision, Recall, RocAucManySlow
from revisit_bpr.models import BPR
from revisit_bpr.models.bpr import MF
from revisit_bpr.modules import AdaptiveSampler

log = logging.getLogger("example")


def build_metrics() -> dict:
    out = {}
    for k in (100, 10, 5, 50):
        out[f"ndcg@{k}"] = NDCG(topk=k)
        out[f"recall@{k}"] = Recall(topk=k)
    out["recall@20"] = Recall(topk=20)
    for k in (5, 10, 50, 100):
        out[f"precision@{k}"] = Precision(topk=k)
    out["auc"] = RocAucManySlow()
    return out


def strict_epoch(model, optimizer, sampler, users, items, seen_pad, batch_size, generator):
    """The reference's train_one_epoch (example.py:157-192) verbatim in structure."""
    perm = torch.randperm(users.numel(), device=users.device, generator=generator)
    total, steps = 0.0, 0
    for lo in range(0, perm.numel(), batch_size):
        idx = perm[lo:lo + batch_size]
        batch = {"user": users[idx].long(), "item": items[idx].long().unsqueeze(-1)}
        if seen_pad is not None:
            batch["seen_items"] = seen_pad[batch["user"]]
        batch["neg"] = sampler.sample(batch)
        out = model(batch)
        out["loss"].backward()
        optimizer.step()
        optimizer.zero_grad()
        total += float(out["loss"].detach())
        steps += 1
    return {"loss": total / max(steps, 1)}


@click.command(context_settings={"help_option_names": ["-h", "--help"]})
@click.argument("dataset_path", required=False,
                type=click.Path(exists=True, dir_okay=True, file_okay=False, path_type=Path))
@click.option("--synthetic", "synthetic_name", default=None,
              help="generate a synthetic dataset of this shape instead of reading files")
@click.option("--num-users", type=int, default=136678, show_default=True)
@click.option("--num-items", type=int, default=20109, show_default=True)
@click.option("--embedding-dim", type=int, default=1024, show_default=True)
@click.option("--batch-size", type=int, default=256, show_default=True)
@click.option("--epochs", type=int, default=72, show_default=True)
@click.option("--seed", type=int, default=13, show_default=True)
@click.option("--lr", type=float, default=0.00943667980759196, show_default=True)
@click.option("--sampling-prob", type=float, default=1 / 700, show_default=True)
@click.option("--mode", type=click.Choice(["stream", "batched", "strict"]), default="stream", show_default=True,
              help="stream: fused SGD launches (StreamTrainer); batched: one launch per refresh period with "
                   "virtual mini-batches, any --optimizer (BatchedStreamTrainer); strict: the reference's "
                   "batch loop")
@click.option("--optimizer", "opt_name", type=click.Choice(["sgd", "adam", "rmsprop", "nesterov"]),
              default="sgd", show_default=True, help="batched / strict modes (stream is plain SGD)")
@click.option("--refresh-lag", type=float, default=0.0, show_default=True,
              help="stream mode: 1 = the adaptive snapshot is sorted beside the previous launch "
                   "(StreamTrainer refresh_lag); 0 = the reference's schedule; -1 = by shape and learning rate "
                   "(fast.auto_schedule: lag 1 only inside the staleness budget lr x 2 x launch <= 2,000)")
@click.option("--refresh-cus", type=int, default=64, show_default=True,
              help="stream mode with --refresh-lag > 0: CUs the snapshot sort is masked to")
def main(dataset_path, synthetic_name, num_users, num_items, embedding_dim, batch_size, epochs,
         seed, lr, sampling_prob, mode, opt_name, refresh_lag, refresh_cus):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s | %(message)s")
    if not torch.cuda.is_available():
        raise SystemExit("example.py needs an MI355X (no CPU path)")
    import os

    import torch.distributed as dist

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("BPR_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    if synthetic_name:
        data = synthetic.generate_named(synthetic_name, eval_users=10_000, seed=seed)
        num_users, num_items = data.num_users, data.num_items
    elif dataset_path is not None:
        data = interactions.load_dataset(dataset_path, num_users, num_items)
    else:
        raise SystemExit("give DATASET_PATH or --synthetic NAME")
    torch.manual_seed(seed)
    model = BPR(
        fuse_forward=True,
        logits_model=MF(torch.nn.Embedding(num_users, embedding_dim, padding_idx=0),
                        torch.nn.Embedding(num_items, embedding_dim, padding_idx=0)),
        reg_alphas={"user": 0.0016, "item": 0.0001, "neg": 0.00375},
    ).to(dev)
    t = {k: torch.from_numpy(getattr(data, k)).to(dev)
         for k in ("users", "items", "indptr", "indices", "eval_users", "eval_indptr", "eval_items")}
    metrics = build_metrics()
    every = int(num_items * math.log(num_items) / batch_size)

    def make_optimizer():
        p = model.parameters()
        return {"sgd": lambda: torch.optim.SGD(p, lr=lr),
                "adam": lambda: torch.optim.Adam(p, lr=lr, betas=(0.1, 0.999)),  # ada-sampling-adam.yaml.j2:175
                "rmsprop": lambda: torch.optim.RMSprop(p, lr=lr, alpha=0.9),
                "nesterov": lambda: torch.optim.SGD(p, lr=lr, momentum=0.9, nesterov=True)}[opt_name]()

    if mode == "stream" and opt_name != "sgd":
        raise SystemExit("--mode stream is the fused SGD kernel; use --mode batched for --optimizer " + opt_name)
    if mode == "batched":
        from revisit_bpr.fast import BatchedStreamTrainer

        users_t, items_t, sync = t["users"], t["items"], None
        if world > 1:
            from revisit_bpr.distributed import ItemSync, balanced_user_shards, owner_of

            bounds = balanced_user_shards(data.indptr, world)
            mine = torch.from_numpy(owner_of(data.users, bounds) == rank).to(dev)
            users_t, items_t = users_t[mine].contiguous(), items_t[mine].contiguous()
            f = model.logits_model.get_features()
            sync = ItemSync([f["item"].data] + ([f["item_bias"].data] if f["item_bias"] is not None else []))
        trainer = BatchedStreamTrainer(model, make_optimizer(), users_t, items_t, t["indptr"], t["indices"],
                                       sampler="adaptive", adaptive_p=sampling_prob, batch_size=batch_size,
                                       seed=seed, rank=rank, item_sync=sync)
        run_epoch = trainer.train_epoch
    elif mode == "stream":
        users_t, items_t, sync = t["users"], t["items"], None
        if world > 1:  # this rank trains the triples of its own user range only
            from revisit_bpr.distributed import ItemSync, balanced_user_shards, owner_of

            bounds = balanced_user_shards(data.indptr, world)
            mine = torch.from_numpy(owner_of(data.users, bounds) == rank).to(dev)
            users_t, items_t = users_t[mine].contiguous(), items_t[mine].contiguous()
            f = model.logits_model.get_features()
            sync = ItemSync([f["item"].data] + ([f["item_bias"].data] if f["item_bias"] is not None else []))
        trainer = StreamTrainer(model, users_t, items_t, t["indptr"], t["indices"], lr=lr,
                                sampler="adaptive", adaptive_p=sampling_prob,
                                batch_size=batch_size, seed=seed, rank=rank, item_sync=sync,
                                refresh_lag="auto" if refresh_lag < 0 else refresh_lag,
                                refresh_cus=refresh_cus if refresh_lag > 0 else 0)
        run_epoch = trainer.train_epoch
    elif world > 1:  # reference mini-batches per user shard, item table reconciled by ItemSync
        from revisit_bpr.distributed import ItemSync, balanced_user_shards, owner_of
        from revisit_bpr.fast import StrictTrainer

        bounds = balanced_user_shards(data.indptr, world)
        mine = torch.from_numpy(owner_of(data.users, bounds) == rank).to(dev)
        f = model.logits_model.get_features()
        sync = ItemSync([f["item"].data] + ([f["item_bias"].data] if f["item_bias"] is not None else []))
        optimizer = make_optimizer()
        trainer = StrictTrainer(model, optimizer, t["users"][mine].contiguous(),
                                t["items"][mine].contiguous(), t["indptr"], t["indices"],
                                sampler="adaptive", adaptive_p=sampling_prob, batch_size=batch_size,
                                seed=seed, rank=rank, item_sync=sync)
        run_epoch = trainer.train_epoch
    else:
        model.bind_seen_csr(t["indptr"], t["indices"])
        optimizer = make_optimizer()
        gen = torch.Generator(device=dev).manual_seed(seed)
        sampler = AdaptiveSampler(model, num_items=num_items, sampling_prob=sampling_prob,
                                  every=every, neg_gen=gen)
        sampler.update_stats()
        model.train()

        def run_epoch():
            return strict_epoch(model, optimizer, sampler, t["users"], t["items"], None,
                                batch_size, gen)

    for epoch in range(epochs):
        t0 = time.perf_counter()
        stats = run_epoch()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        log.info("epoch %d | %s | %.2fs (%.1f M triples/s)", epoch + 1,
                 " ".join(f"{k}={v:.4f}" for k, v in stats.items() if isinstance(v, float)), dt,
                 data.nnz / dt / 1e6)
        if world > 1:  # user rows live on their owner: gather them on every rank for the eval
            P = model.logits_model.get_features()["user"].data
            for r in range(world):
                lo_r, hi_r = int(bounds[r]), int(bounds[r + 1])
                if hi_r > lo_r:
                    dist.broadcast(P[lo_r:hi_r], src=r)
        if mode != "stream":
            model.sync()  # lazily-updated rows -> "now" before the tables are read
        if t["eval_users"].numel() and rank == 0:
            model.eval()
            f = model.logits_model.get_features()
            res = evaluate(f["user"].detach(), f["item"].detach(), f["item_bias"], t["eval_users"],
                           t["eval_indptr"], t["eval_items"], t["indptr"], t["indices"], metrics)
            model.train()
            for name in sorted(res, key=lambda x: (len(x), x)):
                log.info("%-14s | %.4f", name, res[name])
        log.info("Finished epoch: %d", epoch + 1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    main()
