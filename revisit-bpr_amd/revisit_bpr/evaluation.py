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


@torch.no_grad()
def evaluate_topk(P: torch.Tensor, Q: torch.Tensor, item_bias, eval_users: torch.Tensor,
                  eval_indptr: torch.Tensor, eval_items: torch.Tensor, seen_indptr: torch.Tensor,
                  seen_indices: torch.Tensor, ks=(5, 10, 20, 50, 100), block: int = 8192,
                  auc: bool = False) -> dict:
    """NDCG / Recall / Precision at every k in `ks` from ONE top-max(ks) per block of users (the
    reference runs a full argsort of I scores per metric object: 14 sorts per batch,
    experiments/bpr/exp.py:369-374 + metrics/metric.py:110-113).  Same values as the metric classes
    (tests/test_gpu_api.py::test_evaluate_topk_equals_metric_classes).  `auc=True` adds the
    ROC-AUC of the reference's RocAucMany (metrics/auc.py:70-130: all positive / negative pairs of a
    row) from the same block of scores: on a ROCm device `bpr_auc_rows` (csrc/bpr_eval.hip, r6: one pass over the
    scores, the positives ranked in LDS), else by rank sums after one sort per block — instead of the [B, I, I]
    comparison."""
    from revisit_bpr.metrics.auc import RocAucManySlow

    auc_metric = RocAucManySlow() if auc else None
    auc_sum = torch.zeros((), device=P.device, dtype=torch.float32)
    lib = None
    if auc and P.is_cuda:  # r6: bpr_auc_rows — one pass over the scores instead of a sort per row
        from revisit_bpr import native

        lib = native.load()
    dev = P.device
    I = Q.shape[0]
    kmax = min(max(ks), I)
    disc = 1.0 / torch.log2(torch.arange(kmax, dtype=torch.float, device=dev) + 2.0)
    sums = {f"{m}@{k}": torch.zeros((), device=dev, dtype=torch.float64)
            for k in ks for m in ("ndcg", "recall", "precision")}
    E = eval_users.numel()
    for lo in range(0, E, block):
        hi = min(lo + block, E)
        users = eval_users[lo:hi].long()
        n = hi - lo
        rows = torch.arange(n, device=dev)
        logits = P[users] @ Q.T
        if item_bias is not None:
            logits += item_bias
        s_lo, s_hi = seen_indptr[users], seen_indptr[users + 1]
        s_cnt = s_hi - s_lo
        tot = int(s_cnt.sum())
        if tot > 0:
            r = torch.repeat_interleave(rows, s_cnt)
            offs = torch.arange(tot, device=dev) - torch.repeat_interleave(
                torch.cumsum(s_cnt, 0) - s_cnt, s_cnt)
            logits[r, seen_indices[torch.repeat_interleave(s_lo, s_cnt) + offs].long()] = -1e13
        logits[:, 0] = -1e13
        t_lo, t_hi = eval_indptr[lo:hi], eval_indptr[lo + 1:hi + 1]
        t_cnt = t_hi - t_lo
        target = torch.zeros(n, I, device=dev, dtype=torch.bool)  # (1 byte per score; only the top-k look-up reads it)
        if int(t_cnt.sum()) > 0:
            target[torch.repeat_interleave(rows, t_cnt),
                   eval_items[int(t_lo[0]):int(t_hi[-1])].long()] = True
        if auc_metric is not None and lib is None:
            auc_metric(logits, target.float())
        elif auc_metric is not None:
            ptr = (eval_indptr[lo:hi + 1] - eval_indptr[lo]).to(torch.int64).contiguous()
            items = eval_items[int(t_lo[0]):int(t_hi[-1])].to(torch.int32).contiguous()
            rows_auc = torch.empty(n, device=dev, dtype=torch.float32)
            native.check(lib.bpr_auc_rows(logits.data_ptr(), n, I, ptr.data_ptr(), items.data_ptr(), rows_auc.data_ptr(),
                                          torch.cuda.current_stream(dev).cuda_stream))
            many = t_cnt > 4096  # (rows the kernel leaves to the sort-based form)
            if bool(many.any()):
                rows_auc[many] = auc_metric.compute(logits[many], target[many].float())
            auc_sum += rows_auc.sum()
        rel = torch.gather(target, 1, torch.topk(logits, kmax, dim=1).indices).float()  # [n, kmax]
        n_pos = t_cnt.float()
        gains = rel * disc
        for k in ks:
            kk = min(k, I)
            hits = rel[:, :kk].sum(1)
            ideal = torch.cumsum(disc, 0)[(n_pos.clamp(max=kk).long() - 1).clamp(min=0)]
            ideal = torch.where(n_pos > 0, ideal, torch.zeros_like(ideal))
            sums[f"ndcg@{k}"] += torch.nan_to_num(gains[:, :kk].sum(1) / ideal).double().sum()
            sums[f"recall@{k}"] += torch.nan_to_num(hits / n_pos).double().sum()
            sums[f"precision@{k}"] += (hits / kk).double().sum()
    out = {k: float(v / max(E, 1)) for k, v in sums.items()}
    if auc_metric is not None:
        out["auc"] = float(auc_metric.get_metric()) if lib is None else float(auc_sum / max(E, 1))
    return out
