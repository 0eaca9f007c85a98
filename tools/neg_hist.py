#!/usr/bin/env python
"""Where do the STREAM kernel's item-row atomics land?  

Trains the bench workload (ML-20M shape, d=128, adaptive p=0.01) for a few epochs, then records the
negatives of one chunk and reports (a) how concentrated positives and negatives are, (b) how much
of the atomic traffic the hot block (top-H by POSITIVE count, as bpr_plan_epoch builds it) catches
vs a hot block chosen by measured positive+negative load, (c) the per-channel load imbalance of
the rows outside the hot block (memory is interleaved over 128 channels in 256-B units), and
(d) the rate of duplicate item rows among triples in flight together (same wave = 2 triples,
same workgroup = 8, whole chip = ~16 k) — what an LDS / ballot combine could remove.

    python tools/neg_hist.py [--epochs 3] > profiles/r03_neg_hist.txt
"""
import argparse
import math
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "revisit-bpr_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=float, default=2.0)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--lr", type=float, default=0.001)
    ap.add_argument("--p", type=float, default=0.01)
    args = ap.parse_args()
    from revisit_bpr import engine as eng
    from revisit_bpr.datasets import synthetic

    dev = torch.device("cuda")
    data = synthetic.generate_named("ml-20m", eval_users=10_000, seed=13)
    U, I, d = data.num_users, data.num_items, args.dim
    g = torch.Generator().manual_seed(13)
    Q = ((torch.rand(I, d, generator=g) - 0.5) / d)
    P = ((torch.rand(U, d, generator=g) - 0.5) / d)
    P[0] = 0
    Q[0] = 0
    P, Q = P.to(dev), Q.to(dev)
    e = eng.Engine(P, Q)
    e.set_reg(0.0016, 0.0001, 0.00375)
    e.set_optimizer(eng.OPT_SGD, lr=args.lr)
    e.bind_seen_csr(torch.from_numpy(data.indptr).to(dev), torch.from_numpy(data.indices).to(dev))
    chunk = int(I * math.log(I) / 256) * 256
    n_chunks = data.nnz // chunk
    su, si = torch.from_numpy(data.users).to(dev), torch.from_numpy(data.items).to(dev)
    users, items = torch.empty_like(su), torch.empty_like(si)
    e.set_stream_opts(True, 0)
    steps = int(args.epochs * n_chunks)
    neg = torch.zeros(chunk, dtype=torch.int32, device=dev)
    for k in range(steps + 1):
        c = k % n_chunks
        if c == 0:
            e.plan_epoch(su, si, chunk, 13 + k // n_chunks, out=(users, items))
        e.adaptive_refresh()
        lo = c * chunk
        e.train_stream(users[lo:lo + chunk], items[lo:lo + chunk], sampler=eng.NEG_ADAPTIVE,
                       neg=neg if k == steps else None, adaptive_p=args.p, seed=13, offset=k * chunk)
    torch.cuda.synchronize()
    pos_c = items[lo:lo + chunk].cpu().numpy()
    neg_c = neg.cpu().numpy()
    usr_c = users[lo:lo + chunk].cpu().numpy()
    pos_all = np.bincount(data.items, minlength=I).astype(np.float64)  # what the hot block is built from
    hp = np.bincount(pos_c, minlength=I).astype(np.float64)
    hn = np.bincount(neg_c, minlength=I).astype(np.float64)
    print(f"after {steps} chunks ({args.epochs} epochs) of the bench workload; one chunk = {chunk} triples")
    print(f"distinct items: positives {int((hp > 0).sum())}, negatives {int((hn > 0).sum())} of {I - 1}")
    for name, h in (("positives", hp), ("negatives", hn), ("pos+neg", hp + hn)):
        srt = np.sort(h)[::-1]
        cs = np.cumsum(srt) / srt.sum()
        print(f"{name:10s} share of the top 64 / 256 / 1024 / 4096 rows: "
              f"{cs[63]:.3f} / {cs[255]:.3f} / {cs[1023]:.3f} / {cs[4095]:.3f}   max row {srt[0] / srt.sum():.4f}")
    for H in (256, 512, 1024):
        by_pos = np.argsort(-pos_all, kind="stable")[:H]
        by_load = np.argsort(-(hp + hn), kind="stable")[:H]
        tot = (hp + hn).sum()
        print(f"hot block of {H}: by positive count catches {(hp + hn)[by_pos].sum() / tot:.3f} of the item-row "
              f"atomics (pos {hp[by_pos].sum() / hp.sum():.3f}, neg {hn[by_pos].sum() / hn.sum():.3f}); "
              f"by measured load {(hp + hn)[by_load].sum() / tot:.3f}; overlap {len(set(by_pos) & set(by_load))}")
    # channel load of the rows OUTSIDE the hot block: row i of d floats starts at byte i*d*4; 256-B
    # interleave over 128 channels -> a 512-B row covers 2 consecutive channels
    row_bytes = d * 4
    for label, hot in (("by positives", np.argsort(-pos_all, kind="stable")[:256]),
                       ("by load", np.argsort(-(hp + hn), kind="stable")[:256]), ("no hot block", [])):
        load = (hp + hn).copy()
        load[list(hot)] = 0
        ch = np.zeros(128)
        for k in range(max(1, row_bytes // 256)):
            np.add.at(ch, ((np.arange(I) * row_bytes) // 256 + k) % 128, load / max(1, row_bytes // 256))
        print(f"channel load outside the hot block ({label}): max / mean = {ch.max() / ch.mean():.3f}")
    # duplicates among triples in flight together: consecutive windows of the chunk (a wave walks
    # 2 runs, a workgroup 8; the whole chip ~16 k triples at once)
    both = np.stack([pos_c, neg_c], 1)
    for w, what in ((2, "wave (2 triples)"), (8, "workgroup (8 triples)"), (16384, "chip (16 k triples)")):
        m = (len(pos_c) // w) * w
        win = both[:m].reshape(-1, 2 * w)
        srt = np.sort(win, axis=1)
        dup = (srt[:, 1:] == srt[:, :-1]).sum()
        print(f"duplicate item rows inside one {what}: {dup / (2 * m):.4f} of the item-row updates")
    same_user = (usr_c[1:] == usr_c[:-1]).mean()
    print(f"adjacent triples sharing the user: {same_user:.3f}; users in the chunk: {len(np.unique(usr_c))}")


if __name__ == "__main__":
    main()
