#!/usr/bin/env python
"""How does the hot kernel's time depend on the state of the model?  Trains the bench workload (ML-20M shape,
d = 128, SGD, adaptive p = 0.01, the reference's snapshot schedule) and, at checkpoints, times k_stream over one
epoch and looks at what the sampler does: where the negatives land (concentration of the item-row atomics), how
deep the walks go (entries of the column scanned per negative), how concentrated the factor picks are.

    python tools/trained_state_probe.py --lr 0.001 --marks 1,10,30,60,100
"""
import argparse, math, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "revisit-bpr_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lr", type=float, default=0.001)
    ap.add_argument("--marks", default="1,10,30,60,100")
    ap.add_argument("--sampler", default="adaptive")
    ap.add_argument("--hot-rows", type=int, default=None)
    ap.add_argument("--hot-reps", type=int, default=1)
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
    P, Q = P.to(dev), Q.to(dev)
    e = eng.Engine(P, Q)
    e.set_reg(0.0016, 0.0001, 0.00375)
    e.set_optimizer(eng.OPT_SGD, lr=args.lr)
    e.bind_seen_csr(torch.from_numpy(data.indptr).to(dev), torch.from_numpy(data.indices).to(dev))
    if args.hot_rows is not None:
        e.set_hot_rows(args.hot_rows, args.hot_reps)
    chunk = int(I * math.log(I) / 256) * 256
    n_chunks = data.nnz // chunk
    su, si = torch.from_numpy(data.users).to(dev), torch.from_numpy(data.items).to(dev)
    users, items = torch.empty_like(su), torch.empty_like(si)
    e.set_stream_opts(True, 0)
    smp = eng.NEG_ADAPTIVE if args.sampler == "adaptive" else eng.NEG_UNIFORM
    neg = torch.zeros(chunk, dtype=torch.int32, device=dev)
    sc = torch.zeros(4, device=dev)
    marks = [int(x) for x in args.marks.split(",")]
    pos_all = np.bincount(data.items, minlength=I).astype(np.float64)
    hot256 = np.argsort(-pos_all, kind="stable")[:256]
    k = 0
    for ep in range(1, max(marks) + 1):
        timed = ep in marks
        e.plan_epoch(su, si, chunk, 13 + ep, out=(users, items))
        sc.zero_()
        if timed:
            e.timing_enable(1)
        for c in range(n_chunks):
            lo = c * chunk
            if smp == eng.NEG_ADAPTIVE:
                e.adaptive_refresh()
            e.train_stream(users[lo:lo + chunk], items[lo:lo + chunk], sampler=smp, adaptive_p=0.01, seed=13,
                           offset=k * chunk, scalars=sc, neg=neg if (timed and c == n_chunks - 1) else None)
            k += 1
        if not timed:
            continue
        ms, n = e.timing_read()
        e.timing_enable(False)
        loss = float(sc[0] / sc[3])
        hn = np.bincount(neg.cpu().numpy(), minlength=I).astype(np.float64)
        srt = np.sort(hn)[::-1]
        cs = np.cumsum(srt) / srt.sum()
        line = (f"epoch {ep:3d} loss {loss:.4f}  k_stream {ms:.4f} ms ({n} launches)  negatives: distinct {int((hn > 0).sum())} "
                f"top64/256/1024 {cs[63]:.3f}/{cs[255]:.3f}/{cs[1023]:.3f} max row {srt[0] / srt.sum():.4f} "
                f"in the hot block (by positives) {hn[hot256].sum() / hn.sum():.3f}")
        if smp == eng.NEG_ADAPTIVE:
            # walk depth: position of the picked item in its factor's order, from the end the walk started at
            uu = users[lo:lo + chunk][:65536].contiguous()
            e.adaptive_refresh()
            ng, fac, rnk = e.sample_adaptive(uu, 0.01, seed=99, return_draws=True)
            order, sigma = e.adaptive_snapshot()
            inv = torch.empty_like(order)
            ar = torch.arange(I, device=dev, dtype=torch.int32).expand(d, I)
            inv.scatter_(1, order.long(), ar)
            pos_in = inv[fac.long(), ng.long()].long()
            puf = e.P[uu.long(), fac.long()]
            depth = torch.where(puf > 0, pos_in, I - 1 - pos_in).float()
            fc = torch.bincount(fac.long(), minlength=d).float()
            fs = torch.sort(fc, descending=True).values.cumsum(0) / fc.sum()
            q = torch.quantile(depth, torch.tensor([0.5, 0.9, 0.99], device=dev))
            line += (f"  walk depth mean {depth.mean():.0f} p50/p90/p99 {q[0]:.0f}/{q[1]:.0f}/{q[2]:.0f} "
                     f">128: {(depth > 128).float().mean():.3f} >512: {(depth > 512).float().mean():.3f} >2048: {(depth > 2048).float().mean():.4f}"
                     f"  factor picks: top1/top8 share {fs[0]:.3f}/{fs[7]:.3f}  sigma max/median {sigma.max() / sigma.median():.2f}")
        print(line, flush=True)
        # the same chunk three ways (20 launches each): adaptive / its captured negatives given / uniform given
        cu, ci = users[lo:lo + chunk], items[lo:lo + chunk]
        rnd = torch.randint(1, I, (chunk,), device=dev, dtype=torch.int32)
        res = []
        for label, kw in (("adaptive", dict(sampler=smp)), ("given: the captured negatives", dict(sampler=eng.NEG_GIVEN, neg=neg)),
                          ("given: uniform random", dict(sampler=eng.NEG_GIVEN, neg=rnd))):
            e.timing_enable(1)
            for r in range(20):
                e.train_stream(cu, ci, adaptive_p=0.01, seed=77, offset=r * chunk, **kw)
            ms, n = e.timing_read()
            e.timing_enable(False)
            res.append(f"{label} {ms:.4f}")
        print("          same chunk, k_stream ms: " + " | ".join(res), flush=True)


if __name__ == "__main__":
    main()
