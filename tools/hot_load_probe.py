#!/usr/bin/env python
"""r6: which rows should a CU keep in LDS?  Trains the bench workload (ML-20M shape, d = 128, SGD lr 0.001, adaptive
p = 0.01; LDS tier on so that it is quick) to a few checkpoints and, on the last launch of the epoch, counts the
item-row updates per row (positives + the sampler's negatives).  Prints what share of them the L most popular rows
BY POSITIVE COUNT (what bpr_plan_epoch measures once per training set) take, what the L most loaded rows take, and
the row writes a 224-workgroup launch saves with either set: load_r - 224 (1 - exp(-load_r / 224)) per row.

    python tools/hot_load_probe.py --marks 1,10,30
"""
import argparse, math, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "revisit-bpr_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lr", type=float, default=0.001)
    ap.add_argument("--marks", default="1,10,30")
    args = ap.parse_args()
    from revisit_bpr import engine as eng
    from revisit_bpr.datasets import synthetic

    dev = torch.device("cuda")
    data = synthetic.generate_named("ml-20m", eval_users=10_000, seed=13)
    U, I, d = data.num_users, data.num_items, 128
    g = torch.Generator().manual_seed(13)
    Q = ((torch.rand(I, d, generator=g) - 0.5) / d)
    P = ((torch.rand(U, d, generator=g) - 0.5) / d)
    P[0] = 0
    Q[0] = 0
    e = eng.Engine(P.to(dev), Q.to(dev))
    e.set_reg(0.0016, 0.0001, 0.00375)
    e.set_optimizer(eng.OPT_SGD, lr=args.lr)
    e.bind_seen_csr(torch.from_numpy(data.indptr).to(dev), torch.from_numpy(data.indices).to(dev))
    e.set_hot_lds(512)
    chunk = int(I * math.log(I) / 256) * 256
    n_chunks = data.nnz // chunk
    su, si = torch.from_numpy(data.users).to(dev), torch.from_numpy(data.items).to(dev)
    users, items = torch.empty_like(su), torch.empty_like(si)
    e.set_stream_opts(True, 0)
    neg = torch.zeros(chunk, dtype=torch.int32, device=dev)
    marks = [int(x) for x in args.marks.split(",")]
    pos_all = np.bincount(data.items, minlength=I).astype(np.float64)
    by_pos = np.argsort(-pos_all, kind="stable")
    W = 224.0
    k = 0
    for ep in range(1, max(marks) + 1):
        e.plan_epoch(su, si, chunk, 13 + ep, out=(users, items))
        for c in range(n_chunks):
            lo = c * chunk
            e.adaptive_refresh()
            last = ep in marks and c == n_chunks - 1
            e.train_stream(users[lo:lo + chunk], items[lo:lo + chunk], sampler=eng.NEG_ADAPTIVE, adaptive_p=0.01, seed=13,
                           offset=k * chunk, neg=neg if last else None)
            k += 1
        if ep not in marks:
            continue
        hn = np.bincount(neg.cpu().numpy(), minlength=I).astype(np.float64)
        hp = np.bincount(items[lo:lo + chunk].cpu().numpy(), minlength=I).astype(np.float64)
        load = hn + hp
        by_load = np.argsort(-load, kind="stable")
        saved = load - W * (1.0 - np.exp(-load / W))
        line = [f"epoch {ep}: {int(load.sum())} item-row updates; negatives on the top 128/256 rows by positives "
                f"{hn[by_pos[:128]].sum() / hn.sum():.3f}/{hn[by_pos[:256]].sum() / hn.sum():.3f}"]
        for L in (64, 128, 192, 256, 512):
            a, b = by_pos[:L], by_load[:L]
            line.append(f"  L={L:3d}: share of updates by positives {load[a].sum() / load.sum():.3f} by load {load[b].sum() / load.sum():.3f}"
                        f" | row writes saved {saved[a].sum() / load.sum():.3f} vs {saved[b].sum() / load.sum():.3f} of all"
                        f" | overlap {len(set(a) & set(b))}")
        print("\n".join(line), flush=True)


if __name__ == "__main__":
    main()
