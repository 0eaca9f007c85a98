#!/usr/bin/env python
"""Triples/s THROUGH experiments/trainer.py (events, eval engine and all) on a config in the reference's schema:
the per-batch API loop (`--train-mode api`: DataLoader -> sampler -> model(batch) -> backward -> optimizer.step)
against what an unchanged command line gets since r5 (`auto`: whole epochs inside the library when nothing observes
single iterations).  Netflix-shaped synthetic set, d = 64, B = 256, SGD / Adam, 3 epochs each.
    python tools/bench_trainer_path.py [ml-20m 128]     another shape: the library modes only (auto, stream), SGD"""
import sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "revisit-bpr_amd")]
import torch
from click.testing import CliRunner
from experiments import run as run_mod
from revisit_bpr.datasets import interactions, synthetic

CONFIG = ROOT / "tests" / "configs" / "bpr_small.yaml.j2"
NAME = sys.argv[1] if len(sys.argv) > 1 else "netflix"
DIM = int(sys.argv[2]) if len(sys.argv) > 2 else 64
import os
EVAL_USERS = int(os.environ.get("EVAL_USERS", "20"))  # 10000 = the reference's ML-20M protocol (10 k held-out users)
LR = os.environ.get("LR", "")                          # "" = the config's default 0.05; the metric's config: 0.001
FULL = os.environ.get("FULL_METRICS", "")              # 1 = the 14 metrics of the reference's ML-20M config
data = synthetic.generate_named(NAME, eval_users=EVAL_USERS, seed=3)
with tempfile.TemporaryDirectory() as tmp:
    interactions.write_dataset(data, Path(tmp) / "data")
    def run(variant, mode, epochs):
        extra = (f"dataset={tmp}/data;num_users={data.num_users - 1};num_items={data.num_items - 1};"
                 f"embedding_dim={DIM};train_batch_size=256;epochs={epochs};adaptive=1;item_bias=false"
                 + (f";lr={LR}" if LR else "") + (";full_metrics=1" if FULL else ""))
        if variant == "adam":
            extra += ";optimizer=torch.optim.Adam;lr=0.001"
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = CliRunner().invoke(run_mod.main, [str(CONFIG), "--extra-vars", extra, "--train-mode", mode],
                                 catch_exceptions=False, standalone_mode=False)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, res.return_value

    for variant in (("sgd", "adam") if NAME == "netflix" else ("sgd",)):
        for mode in (("api", "auto", "stream") if NAME == "netflix" else ("auto", "strict")):
            run(variant, mode, 1)  # warm (library load, first-launch setup)
            e_hi = 3 if mode == "api" or (NAME != "netflix" and mode == "strict") else 9
            _, exp = run(variant, mode, e_hi)
            # one training epoch + the evaluation after it, from the history's own clock (rows carry perf_counter
            # stamps: dataset loading and model construction stay outside): eval k -> eval k + 1, first epoch left out
            evals = [r for r in exp.history if r["engine"] == "eval"]
            trains = [r for r in exp.history if r["engine"] == "train"]
            per_epoch = (evals[-1]["t"] - evals[1]["t"]) / (len(evals) - 2)
            eval_only = sum(e["t"] - t["t"] for e, t in zip(evals[2:], trains[1:])) / (len(evals) - 2)
            torch.cuda.synchronize()
            print(f"{variant:5s} --train-mode {mode:6s} ({exp._train_mode}{', fused eval' if exp._eval_fused else ''}; {EVAL_USERS} eval users, "
                  f"{len(exp._metrics)} metrics): {per_epoch * 1e3:8.1f} ms per epoch + its evaluation (evaluation {eval_only * 1e3:.1f} ms) "
                  f"= {data.nnz / per_epoch / 1e6:6.2f} M triples/s through Trainer.run; ndcg@100 {evals[0]['ndcg@100']:.3f} -> "
                  f"{evals[-1]['ndcg@100']:.3f}", flush=True)
