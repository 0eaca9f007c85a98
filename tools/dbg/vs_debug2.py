import sys, itertools, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "revisit-bpr_amd"); sys.path.insert(0, "tests")
import oracle
from test_gpu_parity import dev, make_engine, rand_problem
from test_gpu_vstream import OPTS, REG, oracle_batches

def run(opt_name, d, dup, pad, ragged, split, mode="vs"):
    cfg = OPTS[opt_name]
    U, I, B, steps = 300, 200, 24, 60
    P, Q, *_ = rand_problem(U, I, d, 10, seed=d + 3, B=8)
    P *= 4; Q *= 4
    rng = np.random.default_rng(5)
    n = steps * B - (7 if ragged else 0)
    users = rng.integers(1, U, n).astype(np.int32)
    pos = rng.integers(1, I, n).astype(np.int32)
    neg = rng.integers(1, I, n).astype(np.int32)
    if dup:
        users[:6] = users[6]; pos[8:14] = pos[14]; neg[16:20] = pos[14]
    if pad:
        users[40] = 0; pos[41] = 0
    e = make_engine(P, Q, None, REG)
    e.set_optimizer(**cfg); e.alloc_opt_state()
    cut = 20 * B if split else n
    if mode == "vs":
        e.train_stream_batched(dev(users[:cut]), dev(pos[:cut]), B, sampler=0, neg=dev(neg[:cut]), max_inflight=1)
        if split:
            e.train_stream_batched(dev(users[cut:]), dev(pos[cut:]), B, sampler=0, neg=dev(neg[cut:]), max_inflight=1)
    else:
        e.train_strict(dev(users), dev(pos), B, sampler=0, neg=dev(neg))
    e.flush_lazy()
    Po, Qo = P.copy(), Q.copy()
    oracle_batches(Po, Qo, None, users, pos, neg, B, cfg)
    eP, eQ = np.abs(e.P.cpu().numpy() - Po), np.abs(e.Q.cpu().numpy() - Qo)
    return eP.max(), (eP > 2e-5).sum(), eQ.max(), (eQ > 2e-5).sum(), np.unique(np.nonzero(eP > 2e-5)[0])[:6]

for opt_name, d in (("adam_01", 256), ("adam_09", 128)):
    for flags in [(0,0,0,0), (1,0,0,0), (0,1,0,0), (0,0,1,0), (0,0,0,1), (1,1,1,1)]:
        for mode in ("vs", "strict"):
            r = run(opt_name, d, *flags, mode=mode)
            print(opt_name, d, "dup,pad,ragged,split=", flags, mode, "maxP %.2e n %d maxQ %.2e n %d rows %s" % r[:5])
