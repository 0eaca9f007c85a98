"""example.py end to end on a synthetic JSONL dataset directory (reference file layout)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


MODES = {"stream": ["--mode", "stream"], "strict": ["--mode", "strict"],
         "stream-lag": ["--mode", "stream", "--refresh-lag", "1"],
         "batched-adam": ["--mode", "batched", "--optimizer", "adam", "--lr", "0.002"]}


@pytest.mark.parametrize("mode", list(MODES))
def test_example_cli_runs_and_learns(tmp_path, mode, caplog):
    import importlib.util
    import logging
    from pathlib import Path

    from click.testing import CliRunner

    from revisit_bpr.datasets import interactions, synthetic

    data = synthetic.generate_latent(800, 300, 20000, seed=3)
    interactions.write_dataset(data, tmp_path)
    loaded = interactions.load_dataset(tmp_path, data.num_users, data.num_items)
    assert np.array_equal(loaded.indices, data.indices) and np.array_equal(loaded.users, data.users)
    spec = importlib.util.spec_from_file_location(
        "bpr_example", Path(__file__).resolve().parents[1] / "revisit-bpr_amd" / "example.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with caplog.at_level(logging.INFO, logger="example"):
        res = CliRunner().invoke(mod.main, [str(tmp_path), "--num-users", str(data.num_users),
                                            "--num-items", str(data.num_items), "--embedding-dim", "32",
                                            "--epochs", "4", "--lr", "0.05", "--sampling-prob", "0.05",
                                            *MODES[mode]], catch_exceptions=False, standalone_mode=False)
    assert res.exit_code == 0, res.output
    nd = [float(r.getMessage().split("|")[1]) for r in caplog.records if r.getMessage().startswith("ndcg@100")]
    assert len(nd) == 4 and nd[-1] > nd[0] and nd[-1] > 0.08, nd


@pytest.mark.parametrize("mode", list(MODES))
def test_example_two_ranks_on_one_gpu(tmp_path, mode):
    """The multi-GPU path of example.py (user shards + ItemSync) with two ranks sharing cuda:0 over
    gloo — a functional check of sharding, delta all-reduce and the user-row gather; RCCL needs one
    device per rank, the protocol is the same."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    from revisit_bpr.datasets import interactions, synthetic

    data = synthetic.generate_latent(800, 300, 20000, seed=3)
    interactions.write_dataset(data, tmp_path)
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, BPR_DIST_BACKEND="gloo", PYTHONPATH=str(root / "revisit-bpr_amd"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(29611 + list(MODES).index(mode)), str(root / "revisit-bpr_amd" / "example.py"),
           str(tmp_path), "--num-users", str(data.num_users), "--num-items", str(data.num_items),
           "--embedding-dim", "32", "--epochs", "4", "--lr", "0.05", "--sampling-prob", "0.05",
           *MODES[mode]]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    nd = [float(line.rsplit("|", 1)[1]) for line in res.stderr.splitlines() if "ndcg@100" in line]
    assert len(nd) == 4 and nd[-1] > nd[0] and nd[-1] > 0.08, (nd, res.stderr[-1500:])
