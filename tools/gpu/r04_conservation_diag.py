import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/revisit-bpr_amd", "/root/repo/tests"]
import numpy as np, torch
import oracle
from test_gpu_parity import dev, make_engine, rand_problem
from test_gpu_vstream import oracle_batches, oracle_state, distinct_neg, REG
for direct in ("0", "1"):
    os.environ["BPR_VS_DIRECT"] = direct
    for mom in (0.99999,):
        cfg = dict(kind=1, lr=0.0, momentum=mom)
        U, I, B, d, n, launches = 20000, 40, 64, 128, 32768, 3
        P, Q, _, _, _, _, _ = rand_problem(U, I, d, 5, seed=4, B=8)
        rng = np.random.default_rng(8)
        e = make_engine(P, Q, None, REG); e.set_optimizer(**cfg); state = e.alloc_opt_state()
        Po, Qo = P.copy(), Q.copy(); st = oracle_state(Po, Qo, None)
        allu = []
        for t in range(launches):
            users = rng.integers(1, U, n).astype(np.int32); pos = rng.integers(1, I, n).astype(np.int32)
            neg = distinct_neg(pos, rng.integers(1, I, n).astype(np.int32), I)
            e.train_stream_batched(dev(users), dev(pos), B, sampler=0, neg=dev(neg))
            oracle_batches(Po, Qo, None, users, pos, neg, B, cfg, t0=t * (n // B), st=st)
            allu.append(users)
        e.flush_lazy()
        cnt = np.bincount(np.concatenate(allu), minlength=U)
        for k in ("mP", "mQ"):
            got, want = state[k].cpu().numpy(), st[k]
            rs = np.abs(want).max(axis=1)
            err = np.abs(got - want).max(axis=1) / np.maximum(rs, 1e-12)
            badrows = np.nonzero(err > 2e-3)[0]
            print(f"direct={direct} {k}: rows {len(rs)} bad {len(badrows)} max rel err {err.max():.4f} median err {np.median(err):.2e}")
            for r in badrows[:6]:
                j = np.argmax(np.abs(want[r]))
                print("   row", r, "touches", cnt[r] if k == "mP" else -1, "got/want", got[r, j] / want[r, j], "err", err[r])
