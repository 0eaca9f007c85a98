#!/usr/bin/env python
"""Run an experiment config: ``python -m experiments.run CONFIG.yaml.j2 --extra-vars "dataset=...;
num_users=...;num_items=...;embedding_dim=128;train_batch_size=256"`` — jinja2 render → YAML →
instantiate(config["experiment"]) → .run()  (the reference's experiments/run.py:142-187 without
hydra, Optuna, trackers or S3)."""
from __future__ import annotations

import logging
from pathlib import Path

import click

from experiments.config import instantiate, parse_extra_vars, render


@click.command(context_settings={"help_option_names": ["-h", "--help"]})
@click.argument("config_path", type=click.Path(exists=True, dir_okay=False, path_type=Path))
@click.option("--extra-vars", default="", help='template variables, "k=v;k2=v2"')
@click.option("-d", "--dir", "exp_dir", type=click.Path(path_type=Path), default=None)
@click.option("--seed", type=int, default=13, show_default=True)
@click.option("--debug", is_flag=True)
@click.option("--train-mode", type=click.Choice(["auto", "api", "strict", "stream"]), default="auto",
              show_default=True, help="api: the reference's per-batch loop; strict: the same "
              "mini-batches, whole epochs inside the library; stream: the fused SGD throughput path")
def main(config_path: Path, extra_vars: str, exp_dir, seed: int, debug: bool, train_mode: str):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s | %(message)s")
    config = render(config_path, parse_extra_vars(extra_vars))
    exp_cfg = config.pop("experiment")
    extra = {"train_mode": train_mode}
    experiment = instantiate(exp_cfg, exp_config=lambda: config, dir=exp_dir, seed=seed, debug=debug,
                             **extra)
    experiment.run()
    return experiment


if __name__ == "__main__":
    main()
