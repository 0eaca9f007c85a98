"""numpy restatement of the reference's ranking metrics — TEST INFRASTRUCTURE ONLY.

Follows revisit_bpr/metrics/metric.py:110-113 (prepare_target), ndcg.py:8-13,69-78,
recall.py:44-51, precision.py:44-51, auc.py:36-47,149-166 of the reference.  Pinned against the
reference's own outputs in tests/golden/metrics.npz (tests/test_oracle_golden.py).
"""
from __future__ import annotations

import numpy as np


def prepare_target(output: np.ndarray, target: np.ndarray) -> np.ndarray:
    # metric.py:110-113: argsort(-output) then gather target (stable keeps lower ids first on ties)
    idx = np.argsort(-output, axis=-1, kind="stable")
    return np.take_along_axis(target, idx, axis=-1)


def _exp_dcg(t: np.ndarray) -> np.ndarray:
    # ndcg.py:8-13
    gains = (2.0 ** t) - 1.0
    return gains / np.log2(np.arange(t.shape[-1], dtype=np.float32) + 2.0)


def ndcg(output, target, topk: int) -> np.ndarray:
    # ndcg.py:69-78
    k = min(output.shape[-1], topk)
    pred = prepare_target(output, target)[:, :k]
    ideal = prepare_target(target, target)[:, :k]
    with np.errstate(invalid="ignore", divide="ignore"):
        score = _exp_dcg(pred).sum(-1) / _exp_dcg(ideal).sum(-1)
    return np.nan_to_num(score, nan=0.0)


def recall(output, target, topk: int) -> np.ndarray:
    # recall.py:44-51 — hits@k / ALL positives
    k = min(output.shape[-1], topk)
    pred = prepare_target(output, target)[:, :k]
    with np.errstate(invalid="ignore", divide="ignore"):
        score = pred.sum(-1) / target.sum(-1)
    return np.nan_to_num(score, nan=0.0)


def precision(output, target, topk: int) -> np.ndarray:
    # precision.py:44-51
    k = min(output.shape[-1], topk)
    return prepare_target(output, target)[:, :k].sum(-1) / k


def roc_auc_one(output, mask=None) -> np.ndarray:
    # auc.py:36-47: column 0 is the positive
    if mask is None:
        mask = np.ones_like(output)
    m = mask[:, 1:]
    score = (output[:, :1] > output[:, 1:]).astype(np.float32)
    score[m == 0] = 0.0
    return score.sum(-1) / m.sum(-1)


def roc_auc_many(output, target, mask=None) -> np.ndarray:
    # auc.py:149-166 (RocAucManySlow): fraction of (pos, neg) pairs ranked correctly
    if mask is None:
        mask = np.ones_like(output)
    out = np.empty(output.shape[0], np.float32)
    for b in range(output.shape[0]):
        pos = output[b][target[b] != 0]
        neg = output[b][(target[b] == 0) & (mask[b] != 0)]
        with np.errstate(invalid="ignore", divide="ignore"):
            out[b] = (pos[:, None] > neg[None, :]).sum() / np.float32(pos.size * neg.size)
    return out
