"""The PRODUCT metric classes (revisit_bpr.metrics.*) and the fused eval scorer
(revisit_bpr.evaluation.evaluate_topk) against the reference's own values on fixed logits:
tests/golden/metrics.npz, written by tests/golden/make_golden.py from /root/reference
(revisit_bpr/metrics/{ndcg,recall,precision,map,fbeta,auc}.py).  Covers the nan_to_num branch (a user
without positives), I < k, masked (seen) items, the AUC family's mask argument and the streaming
`__call__` / `get_metric` protocol.  Runs on CPU tensors in the CPU suite and on the ROCm device
under `-m gpu` (the eval path of the north-star is PyTorch-ROCm).  Tolerance 1e-6.
"""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "revisit-bpr_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

torch = pytest.importorskip("torch")

KS = (5, 10, 20, 50, 100)
DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


def same(got, want, tol=1e-6):
    got = np.asarray(got.detach().cpu().numpy() if hasattr(got, "detach") else got, np.float64)
    want = np.asarray(want, np.float64)
    if got.shape != want.shape or not np.array_equal(np.isnan(got), np.isnan(want)):
        return False
    ok = ~np.isnan(want)
    return bool(np.all(np.abs(got[ok] - want[ok]) <= tol))


@pytest.fixture(scope="module")
def g():
    return np.load(ROOT / "tests" / "golden" / "metrics.npz")


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("name", ["wide", "narrow"])
def test_topk_metric_classes_match_reference(g, name, device):
    from revisit_bpr.metrics import MAP, NDCG, FBeta, Precision, Recall

    lo = torch.from_numpy(g[f"{name}_logits"]).to(device)
    ta = torch.from_numpy(g[f"{name}_target"]).to(device)
    for k in KS:
        assert same(NDCG(topk=k).compute(lo, ta), g[f"{name}_ndcg@{k}"]), ("ndcg", k)
        assert same(Recall(topk=k).compute(lo, ta), g[f"{name}_recall@{k}"]), ("recall", k)
        assert same(Precision(topk=k).compute(lo, ta), g[f"{name}_precision@{k}"]), ("precision", k)
        assert same(MAP(topk=k).compute(lo, ta), g[f"{name}_map@{k}"]), ("map", k)
        assert same(MAP(topk=k, normalized=False).compute(lo, ta), g[f"{name}_map_raw@{k}"]), ("map raw", k)
        assert same(FBeta(topk=k).compute(lo, ta), g[f"{name}_f1@{k}"]), ("f1", k)
        assert same(FBeta(topk=k, beta=0.5).compute(lo, ta), g[f"{name}_f0.5@{k}"]), ("f0.5", k)
        assert same(NDCG(topk=k, gain_function="linear").compute(lo, ta), g[f"{name}_ndcg_linear@{k}"]), k
    # the streaming protocol: two calls, one ratio (metric.py Metric.__call__ / get_metric)
    for cls, key in ((NDCG, "ndcg"), (MAP, "map"), (FBeta, "f1"), (Recall, "recall"), (Precision, "precision")):
        m = cls(topk=10)
        m.reset()
        m(lo, ta)
        m(lo[:3], ta[:3])
        assert same(m.get_metric(), g[f"{name}_{key}@10_stream"]), key


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("name", ["wide", "narrow"])
def test_auc_classes_match_reference(g, name, device):
    from revisit_bpr.metrics import RocAucMany, RocAucManySlow, RocAucOne

    lo = torch.from_numpy(g[f"{name}_logits"]).to(device)
    ta = torch.from_numpy(g[f"{name}_target"]).to(device)
    mk = torch.from_numpy(g[f"{name}_mask"]).to(device)
    assert same(RocAucManySlow().compute(lo, ta), g[f"{name}_auc_many"])
    assert same(RocAucManySlow().compute(lo, ta, mk), g[f"{name}_auc_many_masked"])
    assert same(RocAucMany().compute(lo, ta), g[f"{name}_auc_many_dense"])
    assert same(RocAucMany().compute(lo, ta, mk), g[f"{name}_auc_many_dense_masked"])
    assert same(RocAucOne().compute(lo, ta), g[f"{name}_auc_one"])
    assert same(RocAucOne().compute(lo, ta, mk), g[f"{name}_auc_one_masked"])
    m = RocAucManySlow()
    m.reset()
    m(lo[1:], ta[1:], mk[1:])
    m(lo[1:4], ta[1:4])
    assert same(m.get_metric(), g[f"{name}_auc_many_stream"])


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("name", ["wide", "narrow"])
def test_evaluate_topk_matches_reference_metric_values(g, name, device):
    """The fused scorer on the same scores: user u's embedding is the unit vector e_u and item i's
    embedding is column i of the fixture's logits, so P Qᵀ IS the fixture (exactly: one product by
    1.0 and zeros); the scorer masks item 0 itself and — for the fixture's row with masked items —
    through the seen CSR.  Its means over users must be the means of the reference's per-user
    values (row 0, the user without positives, contributes 0 to NDCG / Recall and is left out of
    the AUC, whose reference value there is 0/0)."""
    from revisit_bpr.evaluation import evaluate_topk

    lo, ta = g[f"{name}_logits"].copy(), g[f"{name}_target"].copy()
    nb, ni = lo.shape
    masked = np.nonzero(lo[2] <= -1e12)[0]
    masked = masked[masked != 0]
    lo[2, masked] = 0.25  # unmasked scores: the scorer has to mask them through the seen CSR
    P = torch.zeros(nb + 1, nb, device=device)
    P[torch.arange(1, nb + 1), torch.arange(nb)] = 1.0  # users are 1..nb (0 = pad)
    Q = torch.from_numpy(np.ascontiguousarray(lo.T)).to(device)  # [ni, nb]
    ta0 = ta.copy()
    ta0[:, 0] = 0.0
    eval_users = torch.arange(1, nb + 1, device=device, dtype=torch.int32)
    cols = [np.nonzero(ta0[r])[0] for r in range(nb)]
    eval_indptr = torch.from_numpy(np.concatenate([[0], np.cumsum([len(c) for c in cols])])).to(device)
    eval_items = torch.from_numpy(np.concatenate(cols).astype(np.int32)).to(device)
    seen_len = np.zeros(nb + 2, np.int64)
    seen_len[2 + 1 + 1] = len(masked)  # user id 3 = fixture row 2
    seen_indptr = torch.from_numpy(np.cumsum(seen_len)).to(device)[: nb + 2]
    seen_indices = torch.from_numpy(masked.astype(np.int32)).to(device)
    for block in (4096, 4):
        out = evaluate_topk(P, Q, None, eval_users, eval_indptr, eval_items, seen_indptr, seen_indices,
                            ks=KS, block=block, auc=False)
        for k in KS:
            for m in ("ndcg", "recall", "precision"):
                want = float(np.mean(g[f"{name}_pad0_{m}@{k}"].astype(np.float64)))
                assert abs(out[f"{m}@{k}"] - want) <= 1e-6, (m, k, out[f"{m}@{k}"], want)
    # AUC over the users that have positives
    keep = np.nonzero(ta0.sum(1) > 0)[0]
    eu = eval_users[torch.from_numpy(keep).to(device)]
    ip = torch.from_numpy(np.concatenate([[0], np.cumsum([len(cols[r]) for r in keep])])).to(device)
    it = torch.from_numpy(np.concatenate([cols[r] for r in keep]).astype(np.int32)).to(device)
    out = evaluate_topk(P, Q, None, eu, ip, it, seen_indptr, seen_indices, ks=(10,), auc=True)
    want = float(np.mean(g[f"{name}_pad0_auc_many"][keep].astype(np.float64)))
    assert abs(out["auc"] - want) <= 1e-6, (out["auc"], want)
