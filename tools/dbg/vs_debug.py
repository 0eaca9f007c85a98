import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "revisit-bpr_amd"); sys.path.insert(0, "tests")
import oracle
from test_gpu_parity import dev, make_engine, rand_problem
from test_gpu_vstream import OPTS, REG, oracle_batches

def run(opt_name, d, mode, steps=60, B=24, seed_extra=3):
    cfg = OPTS[opt_name]
    U, I = 300, 200
    P, Q, *_ = rand_problem(U, I, d, 10, seed=d + seed_extra, B=8)
    P *= 4; Q *= 4
    rng = np.random.default_rng(5)
    n = steps * B
    users = rng.integers(1, U, n).astype(np.int32)
    pos = rng.integers(1, I, n).astype(np.int32)
    neg = rng.integers(1, I, n).astype(np.int32)
    e = make_engine(P, Q, None, REG)
    e.set_optimizer(**cfg); e.alloc_opt_state()
    if mode == "vs":
        e.train_stream_batched(dev(users), dev(pos), B, sampler=0, neg=dev(neg), max_inflight=1)
    else:
        e.train_strict(dev(users), dev(pos), B, sampler=0, neg=dev(neg))
    e.flush_lazy()
    Po, Qo = P.copy(), Q.copy()
    oracle_batches(Po, Qo, None, users, pos, neg, B, cfg)
    return e.P.cpu().numpy(), e.Q.cpu().numpy(), Po, Qo

for opt_name in ("adam_01", "adam_09", "rmsprop", "sgd"):
    for d in (50, 128, 256):
        out = {}
        for mode in ("vs", "strict"):
            Pg, Qg, Po, Qo = run(opt_name, d, mode)
            eP, eQ = np.abs(Pg - Po), np.abs(Qg - Qo)
            out[mode] = (Pg, Qg)
            print(f"{opt_name:8s} d={d:4d} {mode:6s} vs oracle: maxP {eP.max():.2e} (#>2e-5: {(eP>2e-5).sum()}/{eP.size}) "
                  f"maxQ {eQ.max():.2e} (#>2e-5: {(eQ>2e-5).sum()})  rows bad P {np.unique(np.nonzero(eP>2e-5)[0])[:8]}")
        dP = np.abs(out["vs"][0] - out["strict"][0]); dQ = np.abs(out["vs"][1] - out["strict"][1])
        print(f"{opt_name:8s} d={d:4d} vs-vs-strict: maxP {dP.max():.2e} #>2e-5 {(dP>2e-5).sum()} maxQ {dQ.max():.2e} #>2e-5 {(dQ>2e-5).sum()}")
