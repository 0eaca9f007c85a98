"""`python -m experiments.run CONFIG --extra-vars ...` on a config in the reference's schema:
jinja2 → YAML → instantiate → BPRExperiment → Trainer → HIP engine."""
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

CONFIG = Path(__file__).parent / "configs" / "bpr_small.yaml.j2"


@pytest.mark.parametrize("variant,mode", [("uniform-sgd-bias", "api"), ("adaptive-adam", "api"),
                                          ("uniform-sgd-bias", "strict"), ("adaptive-adam", "strict"),
                                          ("uniform-sgd-bias", "stream"), ("adaptive-sgd", "stream"),
                                          ("adaptive-adam", "stream"), ("popularity-sgd", "api"),
                                          ("popularity-sgd", "stream"), ("adaptive-adam", "auto"),
                                          ("uniform-sgd-bias", "auto")])
def test_config_run_learns(tmp_path, variant, mode):
    from click.testing import CliRunner

    from experiments import run as run_mod
    from revisit_bpr.datasets import interactions, synthetic

    data = synthetic.generate_latent(900, 320, 24000, seed=6)
    interactions.write_dataset(data, tmp_path / "data")
    extra = (f"dataset={tmp_path / 'data'};num_users={data.num_users - 1};num_items={data.num_items - 1};"
             "embedding_dim=32;train_batch_size=256;epochs=4")
    if variant == "adaptive-adam":
        extra += ";adaptive=1;optimizer=torch.optim.Adam;lr=0.01;item_bias=false"
    if variant == "adaptive-sgd":
        extra += ";adaptive=1;item_bias=false"
    if variant == "popularity-sgd":
        # item_counts + neg_sampling_alpha (reference experiments/bpr/exp.py:85-91): negatives drawn
        # with probability proportional to count ** alpha over the unseen items
        import json

        import numpy as np

        cnt = np.bincount(data.items, minlength=data.num_items)
        with open(tmp_path / "item-counts.jsonl", "w") as f:
            for i in range(1, data.num_items):
                if cnt[i]:
                    f.write(json.dumps({"item": i, "count": int(cnt[i])}) + "\n")
        extra += f";item_counts={tmp_path / 'item-counts.jsonl'};neg_sampling_alpha=0.75;item_bias=false"
    res = CliRunner().invoke(run_mod.main, [str(CONFIG), "--extra-vars", extra, "-d", str(tmp_path / "exp"),
                                            "--train-mode", mode],
                             catch_exceptions=False, standalone_mode=False)
    assert res.exit_code == 0
    exp = res.return_value
    evals = [r for r in exp.history if r["engine"] == "eval"]
    trains = [r for r in exp.history if r["engine"] == "train"]
    assert len(evals) == 5 and len(trains) == 4  # eval before every epoch + once at the end
    assert evals[-1]["ndcg@100"] > evals[0]["ndcg@100"] + 0.05
    assert 0.5 < evals[-1]["auc"] <= 1.0
    assert trains[-1]["bpr_loss"] < trains[0]["bpr_loss"]
    assert (tmp_path / "exp" / "history.json").exists()
    if mode == "auto":  # nothing observes single iterations here: the epochs ran inside the library
        assert exp._train_mode == "strict"
