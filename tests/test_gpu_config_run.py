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
    if mode == "auto":  # nothing observes single iterations here: the epochs ran inside the library —
        # plain SGD inside the staleness budget on the fused STREAM path (r6), anything else as exact mini-batches
        assert exp._train_mode == ("stream" if variant == "uniform-sgd-bias" else "strict")


def test_fused_evaluation_equals_the_eval_engine_loop(tmp_path):
    """r6: with whole epochs inside the library the evaluation is ONE pass too (`evaluate_topk` instead of the
    eval engine's DataLoader -> model(batch) -> 4 metric objects): the same numbers for the same model — the
    untrained one both runs evaluate before their first epoch, and the per-batch run's final model put through
    the fused pass."""
    from click.testing import CliRunner

    from experiments import run as run_mod
    from experiments.bpr.dataset import AllItemsCollator
    from revisit_bpr.datasets import interactions, synthetic

    data = synthetic.generate_latent(900, 320, 24000, seed=6)
    interactions.write_dataset(data, tmp_path / "data")
    extra = (f"dataset={tmp_path / 'data'};num_users={data.num_users - 1};num_items={data.num_items - 1};"
             "embedding_dim=32;train_batch_size=256;epochs=2")
    runs = {}
    for mode in ("api", "strict"):
        res = CliRunner().invoke(run_mod.main, [str(CONFIG), "--extra-vars", extra, "--train-mode", mode],
                                 catch_exceptions=False, standalone_mode=False)
        runs[mode] = res.return_value
    fused, loop = runs["strict"], runs["api"]
    assert fused._eval_fused and not loop._eval_fused  # one pseudo-batch per evaluation: the fused pass ran
    e_f = [r for r in fused.history if r["engine"] == "eval"]
    e_l = [r for r in loop.history if r["engine"] == "eval"]
    for key in ("ndcg@100", "recall@20", "precision@10", "auc"):
        assert abs(e_f[0][key] - e_l[0][key]) < 2e-6, (key, e_f[0][key], e_l[0][key])  # the same untrained model
    # the same TRAINED model both ways: the api run's model through the fused pass
    import torch as _t

    from revisit_bpr.evaluation import evaluate_topk

    step = fused.trainer.engines["eval"]._process
    fused._model.load_state_dict(loop._model.state_dict())
    _t.cuda.synchronize()

    class _E:
        class state:
            metrics = {}

    step(_E, None)
    for key in ("ndcg@100", "recall@20", "precision@10", "auc"):
        assert abs(float(_E.state.metrics[key]) - e_l[-1][key]) < 2e-6, (key, float(_E.state.metrics[key]), e_l[-1][key])
    assert isinstance(loop._datasets["eval"].collate_fn, AllItemsCollator) and evaluate_topk is not None
