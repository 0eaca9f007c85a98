"""Eval path (PyTorch-ROCm): score every item for a block of users, mask what the user has seen,
feed the streaming metrics.  Value-identical to the reference's eval loop (example.py:195-230;
experiments/bpr/exp.py:369-374): logits = P[users] Qᵀ (+ item bias), seen items and item 0 set to
−1e13, every metric consumes (logits, target).  One dense GEMM per block (rocBLAS/hipBLASLt via
torch — the only MFMA-shaped work in the whole path) instead of a [B, I, d] gather + einsum.
"""
from __future__ import annotations

from typing import Mapping

import torch

from revisit_bpr.metrics import Metric


@torch.no_grad()
def evaluate(P: torch.Tensor, Q: torch.Tensor, item_bias, eval_users: torch.Tensor,
             eval_indptr: torch.Tensor, eval_items: torch.Tensor, seen_indptr: torch.Tensor,
             seen_indices: torch.Tensor, metrics: Mapping[str, Metric], block: int = 2048) -> dict:
    """eval_users [E]; eval_indptr [E+1] / eval_items: held-out targets per eval user (CSR);
    seen_indptr [U+1] / seen_indices: items to mask per user (CSR).  Returns {name: float}."""
    dev = P.device
    I = Q.shape[0]
    for m in metrics.values():
        m.reset()
    E = eval_users.numel()
    for lo in range(0, E, block):
        hi = min(lo + block, E)
        users = eval_users[lo:hi].long()
        logits = P[users] @ Q.T
        if item_bias is not None:
            logits += item_bias
        n = hi - lo
        rows = torch.arange(n, device=dev)
        # targets
        t_lo, t_hi = eval_indptr[lo:hi], eval_indptr[lo + 1:hi + 1]
        t_cnt = (t_hi - t_lo)
        target = torch.zeros(n, I, device=dev)
        if int(t_cnt.sum()) > 0:
            r = torch.repeat_interleave(rows, t_cnt)
            target[r, eval_items[int(t_lo[0]):int(t_hi[-1])].long()] = 1.0
        # seen mask
        s_lo, s_hi = seen_indptr[users], seen_indptr[users + 1]
        s_cnt = s_hi - s_lo
        if int(s_cnt.sum()) > 0:
            r = torch.repeat_interleave(rows, s_cnt)
            offs = torch.arange(int(s_cnt.sum()), device=dev) - torch.repeat_interleave(
                torch.cumsum(s_cnt, 0) - s_cnt, s_cnt)
            cols = seen_indices[(torch.repeat_interleave(s_lo, s_cnt) + offs)].long()
            logits[r, cols] = -1e13
        logits[:, 0] = -1e13
        for m in metrics.values():
            m(logits, target)
    return {k: float(m.get_metric()) for k, m in metrics.items()}
