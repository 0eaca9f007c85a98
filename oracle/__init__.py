"""CPU oracle for the BPR-MF hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker (never as the thing shipped).  The C restatement lives in
``bpr_oracle.c`` (each function cites the reference file:line it follows); this module is the numpy
binding plus a numpy restatement of the ranking metrics (``metrics_np``).

Parity pin: see ``bpr_oracle.h`` — golden vectors generated from the reference itself by
``tests/golden/make_golden.py`` and checked in ``tests/test_oracle_golden.py``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "liboracle.so"

SGD, MOMENTUM, ADAM, RMSPROP = 0, 1, 2, 3
NEG_GIVEN, NEG_UNIFORM, NEG_ADAPTIVE = 0, 1, 2


class OptParams(ctypes.Structure):
    _fields_ = [
        ("kind", ctypes.c_int32),
        ("lr", ctypes.c_float),
        ("momentum", ctypes.c_float),
        ("dampening", ctypes.c_float),
        ("nesterov", ctypes.c_int32),
        ("beta1", ctypes.c_float),
        ("beta2", ctypes.c_float),
        ("eps", ctypes.c_float),
        ("alpha", ctypes.c_float),
    ]


def make_opt(kind: int, lr: float, momentum: float = 0.0, dampening: float = 0.0,
             nesterov: bool = False, betas=(0.9, 0.999), eps: float = 1e-8,
             alpha: float = 0.99) -> OptParams:
    return OptParams(kind, lr, momentum, dampening, int(nesterov), betas[0], betas[1], eps, alpha)


def build(force: bool = False) -> Path:
    """Compile liboracle.so with gcc (seconds)."""
    src = _HERE / "bpr_oracle.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            build()
        _lib = ctypes.CDLL(str(_LIB_PATH))
        _lib.orc_adaptive_pick_literal.restype = ctypes.c_int32
        _lib.orc_adaptive_pick.restype = ctypes.c_int32
        _lib.orc_step.restype = ctypes.c_int
        _lib.orc_step_sgd_sparse.restype = ctypes.c_int
    return _lib


def _p(a, dtype):
    """pointer to a C-contiguous numpy array of the given dtype (None → NULL)."""
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.dtype == dtype and a.flags["C_CONTIGUOUS"], (
        getattr(a, "dtype", None), dtype)
    return a.ctypes.data_as(ctypes.c_void_p)


def _f(a):
    return _p(a, np.float32)


def _i32(a):
    return _p(a, np.int32)


def _i64(a):
    return _p(a, np.int64)


c_f = ctypes.c_float
c_i32 = ctypes.c_int32
c_i64 = ctypes.c_int64
c_u64 = ctypes.c_uint64


def philox4x32_10(ctr, key):
    c = (ctypes.c_uint32 * 4)(*ctr)
    k = (ctypes.c_uint32 * 2)(*key)
    o = (ctypes.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def forward(P, Q, bias, users, pos, neg, alphas=(0.0, 0.0, 0.0)):
    B, d = len(users), P.shape[1]
    lp = np.empty(B, np.float32)
    ln = np.empty(B, np.float32)
    sc = np.zeros(4, np.float64)
    lib().orc_forward(_f(P), _f(Q), _f(bias), c_i32(d), _i32(users), _i32(pos), _i32(neg),
                      c_i64(B), c_f(alphas[0]), c_f(alphas[1]), c_f(alphas[2]), _f(lp), _f(ln),
                      _p(sc, np.float64))
    return lp, ln, sc


def dense_grad(P, Q, bias, users, pos, neg, alphas=(0.0, 0.0, 0.0), pad_user=0, pad_item=0):
    U, d = P.shape
    I = Q.shape[0]
    gP = np.empty_like(P)
    gQ = np.empty_like(Q)
    gb = np.empty(I, np.float32)
    lib().orc_dense_grad(_f(P), _f(Q), _f(bias), c_i64(U), c_i64(I), c_i32(d), _i32(users),
                         _i32(pos), _i32(neg), c_i64(len(users)), c_f(alphas[0]), c_f(alphas[1]),
                         c_f(alphas[2]), c_i32(pad_user), c_i32(pad_item), _f(gP), _f(gQ), _f(gb))
    return gP, gQ, gb


def opt_dense(opt: OptParams, t: int, w, g, m=None, v=None):
    lib().orc_opt_dense(ctypes.byref(opt), c_i64(t), _f(w), _f(g), _f(m), _f(v), c_i64(w.size))


def step(P, Q, bias, users, pos, neg, opt: OptParams, t: int, state=None,
         alphas=(0.0, 0.0, 0.0), pad_user=0, pad_item=0):
    """One reference iteration (dense optimizer), in place.  state = dict(mP,vP,mQ,vQ,mb,vb)."""
    U, d = P.shape
    I = Q.shape[0]
    B = len(users)
    st = state or {}
    lp = np.empty(B, np.float32)
    ln = np.empty(B, np.float32)
    sc = np.zeros(4, np.float64)
    rc = lib().orc_step(_f(P), _f(Q), _f(bias), c_i64(U), c_i64(I), c_i32(d), _i32(users),
                        _i32(pos), _i32(neg), c_i64(B), c_f(alphas[0]), c_f(alphas[1]),
                        c_f(alphas[2]), c_i32(pad_user), c_i32(pad_item), ctypes.byref(opt),
                        c_i64(t), _f(st.get("mP")), _f(st.get("vP")), _f(st.get("mQ")),
                        _f(st.get("vQ")), _f(st.get("mb")), _f(st.get("vb")), _f(lp), _f(ln),
                        _p(sc, np.float64))
    assert rc == 0
    return lp, ln, sc


def step_sgd_sparse(P, Q, bias, users, pos, neg, lr, alphas=(0.0, 0.0, 0.0), pad_user=0,
                    pad_item=0):
    U, d = P.shape
    I = Q.shape[0]
    B = len(users)
    lp = np.empty(B, np.float32)
    ln = np.empty(B, np.float32)
    sc = np.zeros(4, np.float64)
    rc = lib().orc_step_sgd_sparse(_f(P), _f(Q), _f(bias), c_i64(U), c_i64(I), c_i32(d),
                                   _i32(users), _i32(pos), _i32(neg), c_i64(B), c_f(alphas[0]),
                                   c_f(alphas[1]), c_f(alphas[2]), c_i32(pad_user),
                                   c_i32(pad_item), c_f(lr), _f(lp), _f(ln), _p(sc, np.float64))
    assert rc == 0
    return lp, ln, sc


def sampling_weights(base, seen_padded):
    B, S = seen_padded.shape
    I = base.shape[0]
    out = np.empty((B, I), np.float32)
    lib().orc_sampling_weights(_f(base), c_i64(I), _i64(seen_padded), c_i64(B), c_i64(S), _f(out))
    return out


def sample_uniform(indptr, indices, I, users, seed, offset=0):
    out = np.empty(len(users), np.int32)
    lib().orc_sample_uniform(_i64(indptr), _i32(indices), c_i64(I), _i32(users),
                             c_i64(len(users)), c_u64(seed), c_u64(offset), _i32(out))
    return out


def alias_table(weights):
    """Walker / Vose alias table over weights[1:] (weights[0], the pad item, is ignored): returns
    (accept float32 [I], alias int32 [I]) such that drawing a column c uniformly in 1..I-1 and
    keeping it with probability accept[c], else taking alias[c], yields item i with probability
    w_i / sum(w).  The product's host code builds the same table (revisit_bpr.engine.alias_table)."""
    w = np.asarray(weights, np.float64)[1:]
    n = w.shape[0]
    assert n >= 1 and (w >= 0).all() and w.sum() > 0
    p = w * (n / w.sum())
    accept = np.ones(n + 1, np.float32)
    alias = np.arange(n + 1, dtype=np.int32)
    small = [i for i in range(n) if p[i] < 1.0]
    large = [i for i in range(n) if p[i] >= 1.0]
    p = p.copy()
    while small and large:
        s_, l_ = small.pop(), large.pop()
        accept[s_ + 1] = np.float32(p[s_])
        alias[s_ + 1] = l_ + 1
        p[l_] -= 1.0 - p[s_]
        (small if p[l_] < 1.0 else large).append(l_)
    return accept, alias


def sample_weighted(indptr, indices, I, users, seed, offset, accept, alias):
    out = np.empty(len(users), np.int32)
    lib().orc_sample_weighted(_i64(indptr), _i32(indices), c_i64(I), _i32(users), c_i64(len(users)),
                              c_u64(seed), c_u64(offset), _f(accept), _i32(alias), _i32(out))
    return out


def adaptive_stats(Q):
    I, d = Q.shape
    QT = np.empty((d, I), np.float32)
    sigma = np.empty(d, np.float32)
    lib().orc_adaptive_stats(_f(Q), c_i64(I), c_i32(d), _f(QT), _f(sigma))
    return QT, sigma


def adaptive_order(QT):
    d, I = QT.shape
    order = np.empty((d, I), np.int32)
    lib().orc_adaptive_order(_f(QT), c_i64(I), c_i32(d), _i32(order))
    return order


def adaptive_pick_literal(QT, indptr, indices, user, factor, rank):
    return int(lib().orc_adaptive_pick_literal(_f(QT), c_i64(QT.shape[1]), _i64(indptr),
                                               _i32(indices), c_i32(user), c_i32(factor),
                                               c_i32(rank)))


def adaptive_pick(order, indptr, indices, user, factor, rank):
    return int(lib().orc_adaptive_pick(_i32(order), c_i64(order.shape[1]), _i64(indptr),
                                       _i32(indices), c_i32(user), c_i32(factor), c_i32(rank)))


def sample_adaptive(P, sigma, order, indptr, indices, users, p, seed, offset=0):
    B = len(users)
    neg = np.empty(B, np.int32)
    fac = np.empty(B, np.int32)
    rnk = np.empty(B, np.int32)
    lib().orc_sample_adaptive(_f(P), c_i32(P.shape[1]), _f(sigma), _i32(order),
                              c_i64(order.shape[1]), _i64(indptr), _i32(indices), _i32(users),
                              c_i64(B), c_f(p), c_u64(seed), c_u64(offset), _i32(neg), _i32(fac),
                              _i32(rnk))
    return neg, fac, rnk


def train_stream_seq(P, Q, bias, users, pos, neg, sampler, lr, alphas=(0.0, 0.0, 0.0),
                     adaptive_p=0.01, sigma=None, order=None, indptr=None, indices=None, seed=0,
                     offset=0, pad_user=0, pad_item=0):
    U, d = P.shape
    I = Q.shape[0]
    sc = np.zeros(4, np.float64)
    lib().orc_train_stream_seq(_f(P), _f(Q), _f(bias), c_i64(U), c_i64(I), c_i32(d), _i32(users),
                               _i32(pos), _i32(neg), c_i64(len(users)), c_i32(sampler),
                               c_f(adaptive_p), _f(sigma), _i32(order), _i64(indptr),
                               _i32(indices), c_u64(seed), c_u64(offset), c_f(alphas[0]),
                               c_f(alphas[1]), c_f(alphas[2]), c_i32(pad_user), c_i32(pad_item),
                               c_f(lr), _p(sc, np.float64))
    return sc


def num_threads() -> int:
    return 1  # the functions above are a scalar port


def max_threads() -> int:
    return int(lib().orc_max_threads())


def train_batches_omp(P, Q, users, pos, B, sampler, lr, alphas=(0.0, 0.0, 0.0), adaptive_p=0.01, QT=None,
                      sigma=None, order=None, refresh_every=0, indptr=None, indices=None, seed=0, offset=0,
                      pad_user=0, pad_item=0, seconds=0.0, threads=0):
    """Mini-batches of the path with OpenMP over the triples of a batch (SURVEY §8d baseline (a));
    updates P / Q (and the snapshot buffers) in place, returns (triples trained, scalars)."""
    U, d = P.shape
    I = Q.shape[0]
    sc = np.zeros(4, np.float64)
    fn = lib().orc_train_batches_omp
    fn.restype = ctypes.c_int64
    done = fn(_f(P), _f(Q), c_i64(U), c_i64(I), c_i32(d), _i32(users), _i32(pos), c_i64(len(users)), c_i64(B),
              c_i32(sampler), c_f(adaptive_p), _f(QT), _f(sigma), _i32(order), c_i64(refresh_every),
              _i64(indptr), _i32(indices), c_u64(seed), c_u64(offset), c_f(alphas[0]), c_f(alphas[1]),
              c_f(alphas[2]), c_i32(pad_user), c_i32(pad_item), c_f(lr), ctypes.c_double(seconds),
              c_i32(threads), _p(sc, np.float64))
    return int(done), sc


__all__ = [n for n in dir() if not n.startswith("_")]
