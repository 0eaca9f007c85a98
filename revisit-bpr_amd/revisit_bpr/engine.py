"""Torch-facing handle on the HIP BPR engine (libbprcore.so through ``revisit_bpr.native``).

torch is used for device memory and streams only: every tensor handed in stays owned by the
caller, the engine receives ``data_ptr()``s and mutates the embedding tables in place.  All
methods are asynchronous on the current torch stream.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from revisit_bpr import native
from revisit_bpr.native import (MODE_STREAM, MODE_STRICT, NEG_ADAPTIVE, NEG_GIVEN, NEG_UNIFORM,
                                OPT_ADAM, OPT_MOMENTUM, OPT_RMSPROP, OPT_SGD)

__all__ = ["Engine", "MaskedStream", "cu_mask", "resolve_reg_alphas", "MODE_STREAM", "MODE_STRICT", "NEG_ADAPTIVE",
           "NEG_GIVEN", "NEG_UNIFORM", "OPT_ADAM", "OPT_MOMENTUM", "OPT_RMSPROP", "OPT_SGD"]


def resolve_reg_alphas(reg_alphas: Optional[dict]) -> tuple[float, float, float]:
    """(user, item, neg) after the rules of the reference's Model.regularization
    (revisit_bpr/models/bpr/model.py:74-86): `all` overrides; missing user/item → 0; missing neg →
    item."""
    reg = reg_alphas or {}
    all_reg, user, item, neg = reg.get("all"), reg.get("user"), reg.get("item"), reg.get("neg")
    if all(r is None for r in (all_reg, user, item, neg)):
        return 0.0, 0.0, 0.0
    if all_reg is not None:
        user = item = neg = all_reg
    user = user or 0
    item = item or 0
    neg = neg or item
    return float(user), float(item), float(neg)


def alias_table(weights: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """Walker / Vose alias table over weights[1:] (entry 0, the pad item, is ignored): (accept
    float32 [I], alias int32 [I]) for `Engine.bind_item_weights`."""
    import numpy as np

    w = weights.detach().double().cpu().numpy()[1:]
    n = w.shape[0]
    if n < 1 or (w < 0).any() or not w.sum() > 0:
        raise ValueError("item weights must be non-negative with a positive sum")
    p = w * (n / w.sum())
    accept = np.ones(n + 1, np.float32)
    alias = np.arange(n + 1, dtype=np.int32)
    small = [i for i in range(n) if p[i] < 1.0]
    large = [i for i in range(n) if p[i] >= 1.0]
    while small and large:
        s_, l_ = small.pop(), large.pop()
        accept[s_ + 1] = np.float32(p[s_])
        alias[s_ + 1] = l_ + 1
        p[l_] -= 1.0 - p[s_]
        (small if p[l_] < 1.0 else large).append(l_)
    return torch.from_numpy(accept), torch.from_numpy(alias)


def cu_mask(first: int, count: int, total: int = 256) -> list[int]:
    """32-bit words of a CU mask with bits [first, first + count) set.  On MI300-class parts the
    driver deals the mask bits round-robin over the XCDs (bit i -> XCD i % 8), so a contiguous bit
    range takes the same number of CUs from every XCD."""
    if not (0 <= first and 0 < count and first + count <= total):
        raise ValueError("CU range out of bounds")
    words = [0] * ((total + 31) // 32)
    for b in range(first, first + count):
        words[b // 32] |= 1 << (b % 32)
    return words


class MaskedStream:
    """A HIP stream restricted to a set of CUs (hipExtStreamCreateWithCUMask through
    ``bpr_stream_create``); `.torch` is the same stream as a torch.cuda.ExternalStream.  The split
    refresh sorts on one while the STREAM kernel runs on the complementary one."""

    def __init__(self, device: torch.device, mask_words: Optional[list[int]]) -> None:
        self._lib = native.load()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        out = ctypes.c_void_p()
        if mask_words:
            arr = (ctypes.c_uint32 * len(mask_words))(*mask_words)
            native.check(self._lib.bpr_stream_create(idx, arr, len(mask_words), ctypes.byref(out)))
        else:
            native.check(self._lib.bpr_stream_create(idx, None, 0, ctypes.byref(out)))
        self.ptr = out.value
        self.torch = torch.cuda.ExternalStream(self.ptr, device=torch.device("cuda", idx))

    def close(self) -> None:
        if getattr(self, "ptr", None):
            self._lib.bpr_stream_destroy(ctypes.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class Engine:
    """One bpr_ctx bound to a pair of embedding tables resident in HBM."""

    def __init__(self, P: torch.Tensor, Q: torch.Tensor, item_bias: Optional[torch.Tensor] = None,
                 pad_user: Optional[int] = 0, pad_item: Optional[int] = 0) -> None:
        self._lib = native.load()
        if not (P.is_cuda and Q.is_cuda):
            raise RuntimeError("Engine needs the embedding tables on a ROCm device; there is no "
                               "CPU path in libbprcore")
        for name, t in (("P", P), ("Q", Q), ("item_bias", item_bias)):
            if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
                raise ValueError(f"{name} must be a contiguous float32 tensor")
        if P.shape[1] != Q.shape[1]:
            raise ValueError("user and item tables must share the embedding dim")
        self.device = P.device
        self.P, self.Q, self.item_bias = P, Q, item_bias
        self.U, self.d = P.shape
        self.I = Q.shape[0]
        self._ctx = ctypes.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        native.check(self._lib.bpr_ctx_create(ctypes.byref(self._ctx), idx, self._stream()))
        native.check(self._lib.bpr_bind_tables(
            self._ctx, P.data_ptr(), self.U, Q.data_ptr(), self.I, self.d, _ptr(item_bias),
            -1 if pad_user is None else int(pad_user), -1 if pad_item is None else int(pad_item)))
        self._keep: dict[str, object] = {}
        self._scalars = torch.zeros(native.SCALARS, dtype=torch.float32, device=self.device)
        self.opt_kind = OPT_SGD
        self.lds_launches = 0
        # test / measurement aids: the environment is read HERE, once per engine, never by the library
        import os

        seen = os.environ.get("BPR_SEEN") or ("csr" if os.environ.get("BPR_NO_BITMAP") else "")
        if seen:
            self.set_tuning("seen", {"csr": 1, "bitmap": 2, "list": 3}[seen])
        if os.environ.get("BPR_VS_DIRECT") in ("0", "1"):
            self.set_tuning("vs_direct", int(os.environ["BPR_VS_DIRECT"]))
        if os.environ.get("BPR_NO_ADAM_CLOSED"):
            self.set_tuning("adam_closed", 0)
        if os.environ.get("BPR_REFRESH_SUB") in ("1", "2", "4"):
            self.set_tuning("refresh_sub", int(os.environ["BPR_REFRESH_SUB"]))
        if os.environ.get("BPR_HOT_LDS"):  # "rows" or "rows,always"
            v = os.environ["BPR_HOT_LDS"].split(",")
            self.set_hot_lds(int(v[0]), len(v) > 1 and v[1] == "1")
        if os.environ.get("BPR_LDS_TAIL"):
            self.set_tuning("lds_tail", int(os.environ["BPR_LDS_TAIL"]))
        if os.environ.get("BPR_LDS_BLOCK"):
            self.set_tuning("lds_block", int(os.environ["BPR_LDS_BLOCK"]))
        if os.environ.get("BPR_HEAVY_T"):
            native.check(self._lib.bpr_set_heavy_users(self._ctx, int(os.environ["BPR_HEAVY_T"]), 0))

    def snapshot_partial(self) -> bool:
        """True while the snapshot the samplers read is a partial one (``bpr_adaptive_snapshot_partial``)."""
        v = ctypes.c_int32()
        native.check(self._lib.bpr_adaptive_snapshot_partial(self._ctx, ctypes.byref(v)))
        return bool(v.value)

    def set_tuning(self, key: str, value: int) -> None:
        """``bpr_set_tuning``: "seen" 0 auto | 1 csr | 2 bitmap | 3 list; "vs_direct" -1 auto | 0 | 1;
        "refresh_sub" 0 | 1 | 2 | 4; "partial_snapshot" 0 | 1; "partial_target" 1..1024; "binned_sort" 1 | 0."""
        native.check(self._lib.bpr_set_tuning(self._ctx, key.encode(), int(value)))
        self.__dict__.setdefault("_tuning", {})[key] = int(value)

    def tuning(self, key: str, default: int = 0) -> int:
        """What `set_tuning(key, ...)` was last given through this engine (`default`: never set)."""
        return self.__dict__.get("_tuning", {}).get(key, default)

    # ---- plumbing ---------------------------------------------------------------------------
    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _sync_stream(self) -> None:
        native.check(self._lib.bpr_set_stream(self._ctx, self._stream()))

    def close(self, fold: bool = True) -> None:
        """Destroy the ctx.  fold (an explicit call: the tables are alive): a communicator with its hot tier still
        open is closed first — `bpr_comm_destroy` folds the exchange in flight and this rank's uncut deltas into the
        item table; the garbage collector's call passes False and `bpr_ctx_destroy` then writes nothing."""
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            if fold:
                self._lib.bpr_comm_destroy(self._ctx)
            self._lib.bpr_ctx_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close(fold=False)
        except Exception:
            pass

    def _ids(self, t: torch.Tensor, name: str) -> torch.Tensor:
        if not t.is_cuda:
            raise ValueError(f"{name} must live on the GPU")
        t = t.reshape(-1)
        if t.dtype != torch.int32:
            t = t.to(torch.int32)
        return t.contiguous()

    # ---- configuration ----------------------------------------------------------------------
    def bind_seen_csr(self, indptr: torch.Tensor, indices: torch.Tensor) -> None:
        """indptr [U+1] int64, indices [nnz] int32 (sorted per row, no 0, no duplicates)."""
        if indptr.dtype != torch.int64 or indices.dtype != torch.int32:
            raise ValueError("indptr must be int64 and indices int32")
        if indptr.numel() != self.U + 1:
            raise ValueError("indptr must have U+1 entries")
        indptr, indices = indptr.contiguous(), indices.contiguous()
        self._keep["csr"] = (indptr, indices)
        native.check(self._lib.bpr_bind_seen_csr(self._ctx, indptr.data_ptr(), indices.data_ptr()))

    def bind_item_weights(self, weights: Optional[torch.Tensor]) -> None:
        """Item weights of the uniform sampler (count_i ** neg_sampling_alpha of the reference's
        BPRExperiment): [I] non-negative; None = uniform."""
        if weights is None:
            self._keep.pop("item_weights", None)
            native.check(self._lib.bpr_bind_item_weights(self._ctx, None, None))
            return
        if weights.numel() != self.I:
            raise ValueError("item weights must have one entry per item row")
        accept, alias = alias_table(weights)
        accept, alias = accept.to(self.device), alias.to(self.device)
        self._keep["item_weights"] = (accept, alias)
        native.check(self._lib.bpr_bind_item_weights(self._ctx, accept.data_ptr(), alias.data_ptr()))

    def set_reg(self, user: float, item: float, neg: float) -> None:
        native.check(self._lib.bpr_set_reg(self._ctx, user, item, neg))

    def set_optimizer(self, kind: int, lr: float, momentum: float = 0.0, dampening: float = 0.0,
                      nesterov: bool = False, betas=(0.9, 0.999), eps: float = 1e-8,
                      alpha: float = 0.99) -> None:
        prm = native.OptParams(lr, momentum, dampening, int(nesterov), betas[0], betas[1], eps,
                               alpha)
        native.check(self._lib.bpr_set_optimizer(self._ctx, kind, ctypes.byref(prm)))
        self.opt_kind = kind
        self._momentum = momentum

    def bind_opt_state(self, mP=None, vP=None, mQ=None, vQ=None, mb=None, vb=None) -> None:
        self._keep["opt_state"] = (mP, vP, mQ, vQ, mb, vb)
        native.check(self._lib.bpr_bind_opt_state(self._ctx, _ptr(mP), _ptr(vP), _ptr(mQ), _ptr(vQ),
                                                  _ptr(mb), _ptr(vb)))

    def alloc_opt_state(self) -> dict:
        """Zero state tensors of the shapes the current optimizer kind needs, bound to the ctx."""
        z = torch.zeros_like
        need_m = self.opt_kind in (OPT_MOMENTUM, OPT_ADAM) or (
            self.opt_kind == OPT_RMSPROP and getattr(self, "_momentum", 0.0) > 0)
        need_v = self.opt_kind in (OPT_ADAM, OPT_RMSPROP)
        st = {
            "mP": z(self.P) if need_m else None, "vP": z(self.P) if need_v else None,
            "mQ": z(self.Q) if need_m else None, "vQ": z(self.Q) if need_v else None,
            "mb": z(self.item_bias) if need_m and self.item_bias is not None else None,
            "vb": z(self.item_bias) if need_v and self.item_bias is not None else None,
        }
        self.bind_opt_state(**st)
        return st

    # ---- negative sampling ------------------------------------------------------------------
    def sample_uniform(self, users: torch.Tensor, seed: int, offset: int = 0) -> torch.Tensor:
        self._sync_stream()
        users = self._ids(users, "users")
        out = torch.empty_like(users)
        native.check(self._lib.bpr_sample_uniform(self._ctx, users.data_ptr(), users.numel(), seed,
                                                  offset, out.data_ptr()))
        return out

    def adaptive_refresh(self) -> None:
        """AdaptiveSampler.update_stats: snapshot of the item table as of now, in stream order."""
        self._sync_stream()
        native.check(self._lib.bpr_adaptive_refresh(self._ctx))

    def adaptive_refresh_begin(self) -> None:
        """Split refresh, first half: the snapshot's keys are cut from the item table NOW (stream
        order) and sorted on the side stream while this stream keeps working; the samplers keep
        reading the previous snapshot until `adaptive_refresh_commit`."""
        self._sync_stream()
        native.check(self._lib.bpr_adaptive_refresh_begin(self._ctx))

    def adaptive_refresh_commit(self) -> None:
        """Split refresh, second half: launches after this call read the snapshot cut by the last
        `adaptive_refresh_begin` (this stream waits for the side stream's sort)."""
        self._sync_stream()
        native.check(self._lib.bpr_adaptive_refresh_commit(self._ctx))

    def _snapshot_views(self, back: bool) -> tuple[torch.Tensor, torch.Tensor]:
        """The engine's snapshot buffers as tensors (no copy): order [d, I] int32, sigma [d] fp32."""
        o, g = ctypes.c_void_p(), ctypes.c_void_p()
        native.check(self._lib.bpr_adaptive_snapshot_ptrs(self._ctx, int(back), ctypes.byref(o),
                                                          ctypes.byref(g)))

        class _Dev:  # torch reads device memory it does not own through this protocol
            def __init__(self, ptr, shape, typestr):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr,
                                                 "data": (ptr, False), "version": 2}

        order = torch.as_tensor(_Dev(o.value, (self.d, self.I), "<i4"), device=self.device)
        sigma = torch.as_tensor(_Dev(g.value, (self.d,), "<f4"), device=self.device)
        return order, sigma

    def adaptive_refresh_sharded(self, rank: int, world: int, group=None, force: bool = False) -> None:
        """``adaptive_refresh`` with the sort shared by the ranks of a multi-GPU job: this rank
        sorts d / world factors of ITS replica of the item table, an all-gather hands every rank
        every factor's order (10 MB at ML-20M / d = 128) and the same snapshot is published
        everywhere.  Falls back to the full refresh when d does not divide by world.
        force: take the sharded route with ONE rank too (part + all-gather + publish through the process
        group: bench.py --force-dist)."""
        import torch.distributed as dist

        if (world <= 1 and not force) or self.d % max(world, 1) != 0:
            return self.adaptive_refresh()
        self._sync_stream()
        if world <= 1:  # forced: one rank's share is every factor — a full refresh, then the same two
            self.adaptive_refresh()  # all-gathers over the published snapshot (the collective path, one rank)
            order, sigma = self._snapshot_views(back=False)
            dist.all_gather_into_tensor(order.view(-1), order.view(-1).clone(), group=group)
            dist.all_gather_into_tensor(sigma, sigma.clone(), group=group)
            return
        per = self.d // world
        native.check(self._lib.bpr_adaptive_refresh_part(self._ctx, rank * per, (rank + 1) * per))
        order, sigma = self._snapshot_views(back=True)
        mine_o = order[rank * per:(rank + 1) * per].clone()
        mine_s = sigma[rank * per:(rank + 1) * per].clone()
        dist.all_gather_into_tensor(order.view(-1), mine_o.view(-1), group=group)
        dist.all_gather_into_tensor(sigma, mine_s, group=group)
        native.check(self._lib.bpr_adaptive_refresh_publish(self._ctx))

    # ---- multi-GPU inside the library (RCCL behind the C ABI; distributed.ItemSync is the torch twin)
    @staticmethod
    def comm_unique_id() -> bytes:
        """ncclGetUniqueId through the library: rank 0 calls it and ships the 128 bytes to the others."""
        buf = ctypes.create_string_buffer(128)
        native.check(native.load().bpr_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int) -> None:
        """Collective: the ctx gets an RCCL communicator; the reconciliation base is cut from the item
        table as it is now (identical on every rank)."""
        self._sync_stream()
        native.check(self._lib.bpr_comm_init(self._ctx, ctypes.c_char_p(unique_id), rank, world))

    def item_sync(self) -> None:
        """Steady-state reconciliation step (fold the all-reduce in flight, cut the next delta,
        all-reduce it on the communicator's stream): `bpr_item_sync`."""
        self._sync_stream()
        native.check(self._lib.bpr_item_sync(self._ctx))

    def item_sync_finish(self) -> None:
        self._sync_stream()
        native.check(self._lib.bpr_item_sync_finish(self._ctx))

    def item_sync_rebase(self) -> None:
        """After the item table was overwritten in place (checkpoint restore): the reconciliation
        bases are cut again from the tables as they are (identical on every rank)."""
        self._sync_stream()
        native.check(self._lib.bpr_item_sync_rebase(self._ctx))

    def comm_hot_tier(self, items: torch.Tensor, counts: Optional[torch.Tensor] = None) -> None:
        """Two tiers inside the library: the hot set (same list on every rank) whose delta block
        `hot_sync` exchanges after every launch; `item_sync` stays the cold rows' per-period step."""
        import numpy as np

        self._sync_stream()
        it = np.ascontiguousarray(items.detach().cpu().numpy(), dtype=np.int32)
        cn = None if counts is None else np.ascontiguousarray(counts.detach().cpu().numpy(), dtype=np.uint32)
        native.check(self._lib.bpr_comm_hot_tier(
            self._ctx, it.ctypes.data_as(ctypes.c_void_p), int(it.size),
            None if cn is None else cn.ctypes.data_as(ctypes.c_void_p)))

    def hot_sync(self) -> None:
        self._sync_stream()
        native.check(self._lib.bpr_hot_sync(self._ctx))

    def comm_destroy(self) -> None:
        native.check(self._lib.bpr_comm_destroy(self._ctx))

    def refresh_pending(self) -> bool:
        out = ctypes.c_int32(0)
        native.check(self._lib.bpr_adaptive_refresh_pending(self._ctx, ctypes.byref(out)))
        return bool(out.value)

    def set_side_stream(self, stream: Optional["MaskedStream"]) -> None:
        """The stream the split refresh sorts on (None: a plain stream owned by the library)."""
        # (the library synchronises the stream it is about to let go of: the old one must still exist then —
        # the reference kept here is replaced AFTER the call)
        native.check(self._lib.bpr_set_side_stream(self._ctx, None if stream is None else stream.ptr))
        self._keep["side_stream"] = stream

    def sample_adaptive(self, users: torch.Tensor, p: float, seed: int, offset: int = 0,
                        return_draws: bool = False):
        self._sync_stream()
        users = self._ids(users, "users")
        neg = torch.empty_like(users)
        fac = torch.empty_like(users) if return_draws else None
        rnk = torch.empty_like(users) if return_draws else None
        native.check(self._lib.bpr_sample_adaptive(self._ctx, users.data_ptr(), users.numel(), p,
                                                   seed, offset, neg.data_ptr(), _ptr(fac),
                                                   _ptr(rnk)))
        return (neg, fac, rnk) if return_draws else neg

    def adaptive_pick(self, users, factor, rank) -> torch.Tensor:
        self._sync_stream()
        users, factor, rank = (self._ids(users, "users"), self._ids(factor, "factor"),
                               self._ids(rank, "rank"))
        neg = torch.empty_like(users)
        native.check(self._lib.bpr_adaptive_pick(self._ctx, users.data_ptr(), factor.data_ptr(),
                                                 rank.data_ptr(), users.numel(), neg.data_ptr()))
        return neg

    def adaptive_snapshot(self) -> tuple[torch.Tensor, torch.Tensor]:
        self._sync_stream()
        order = torch.empty((self.d, self.I), dtype=torch.int32, device=self.device)
        sigma = torch.empty(self.d, dtype=torch.float32, device=self.device)
        native.check(self._lib.bpr_adaptive_get_snapshot(self._ctx, order.data_ptr(),
                                                         sigma.data_ptr()))
        return order, sigma

    def adaptive_sigma(self) -> torch.Tensor:
        """sigma_f of the current snapshot (AdaptiveSampler._factor_std), [d] fp32."""
        self._sync_stream()
        sigma = torch.empty(self.d, dtype=torch.float32, device=self.device)
        native.check(self._lib.bpr_adaptive_get_snapshot(self._ctx, None, sigma.data_ptr()))
        return sigma

    # ---- hot path ---------------------------------------------------------------------------
    def _outs(self, B: int, want_logits: bool, scalars: Optional[torch.Tensor]):
        lp = torch.empty(B, dtype=torch.float32, device=self.device) if want_logits else None
        ln = torch.empty(B, dtype=torch.float32, device=self.device) if want_logits else None
        if scalars is None:
            scalars = torch.zeros(native.SCALARS, dtype=torch.float32, device=self.device)
        return lp, ln, scalars

    def forward(self, users, pos, neg, scalars=None):
        self._sync_stream()
        users, pos, neg = self._ids(users, "users"), self._ids(pos, "pos"), self._ids(neg, "neg")
        lp, ln, sc = self._outs(users.numel(), True, scalars)
        native.check(self._lib.bpr_forward(self._ctx, users.data_ptr(), pos.data_ptr(),
                                           neg.data_ptr(), users.numel(), lp.data_ptr(),
                                           ln.data_ptr(), sc.data_ptr()))
        return lp, ln, sc

    def forward_grad(self, users, pos, neg, scalars=None):
        self._sync_stream()
        users, pos, neg = self._ids(users, "users"), self._ids(pos, "pos"), self._ids(neg, "neg")
        lp, ln, sc = self._outs(users.numel(), True, scalars)
        native.check(self._lib.bpr_forward_grad(self._ctx, users.data_ptr(), pos.data_ptr(),
                                                neg.data_ptr(), users.numel(), lp.data_ptr(),
                                                ln.data_ptr(), sc.data_ptr()))
        return lp, ln, sc

    def apply(self) -> None:
        self._sync_stream()
        native.check(self._lib.bpr_apply(self._ctx))

    def discard_grad(self) -> None:
        self._sync_stream()
        native.check(self._lib.bpr_discard_grad(self._ctx))

    def get_grad(self):
        self._sync_stream()
        gP, gQ = torch.empty_like(self.P), torch.empty_like(self.Q)
        gb = torch.empty(self.I, dtype=torch.float32, device=self.device)
        native.check(self._lib.bpr_get_grad(self._ctx, gP.data_ptr(), gQ.data_ptr(),
                                            gb.data_ptr()))
        return gP, gQ, gb

    def step(self, users, pos, neg=None, mode: int = MODE_STRICT, sampler: int = NEG_GIVEN,
             adaptive_p: float = 0.01, seed: int = 0, offset: int = 0, scalars=None,
             want_logits: bool = True):
        """One reference iteration (sample → forward → backward → optimizer step)."""
        self._sync_stream()
        users, pos = self._ids(users, "users"), self._ids(pos, "pos")
        if neg is None:
            neg = torch.empty_like(users)
        else:
            neg = self._ids(neg, "neg")
        lp, ln, sc = self._outs(users.numel(), want_logits and mode == MODE_STRICT, scalars)
        native.check(self._lib.bpr_step(self._ctx, users.data_ptr(), pos.data_ptr(),
                                        neg.data_ptr(), users.numel(), mode, sampler, adaptive_p,
                                        seed, offset, _ptr(lp), _ptr(ln), sc.data_ptr()))
        return lp, ln, sc, neg

    def train_stream(self, users, pos, sampler: int = NEG_UNIFORM, neg: Optional[torch.Tensor] = None,
                     adaptive_p: float = 0.01, seed: int = 0, offset: int = 0,
                     max_inflight: int = 0, scalars: Optional[torch.Tensor] = None,
                     cut: bool = False) -> None:
        """STREAM mode over users/pos (int32, on device) in one launch; `scalars` (if given) is
        added to.  cut=True: the launch's epilogue also cuts the keys of the next adaptive snapshot
        (``bpr_train_stream_cut``) — the next `adaptive_refresh_begin` only queues the sort.
        cut="async": that cut runs on the side stream beside the NEXT launch and folds nothing
        (``bpr_train_stream_acut``); call `hot_fold()` before reading the item table's storage."""
        self._sync_stream()
        if users.dtype != torch.int32 or pos.dtype != torch.int32:
            raise ValueError("train_stream takes int32 id tensors (no hidden copies on the hot path)")
        if sampler == NEG_GIVEN and neg is None:
            raise ValueError("sampler NEG_GIVEN needs `neg`")
        fn = (self._lib.bpr_train_stream_acut if cut == "async" else
              self._lib.bpr_train_stream_cut if cut else self._lib.bpr_train_stream)
        native.check(fn(self._ctx, users.data_ptr(), pos.data_ptr(), _ptr(neg), users.numel(),
                        sampler, adaptive_p, seed, offset, max_inflight, _ptr(scalars)))
        self.lds_launches += self._lib.bpr_stream_lds_rows(self._ctx) > 0  # launches the LDS-tier kernel ran (r6)

    def set_bias_tracking(self, on: bool) -> None:
        """item_bias during STREAM launches (``bpr_set_bias_tracking``): on = a launch skips re-reading the
        dense vector while the library knows its own one-item-per-line table is current.  Only for a
        loop that owns the vector between its launches (fast.StreamTrainer switches it on for the
        duration of one epoch call, bench.py for its run): a write through torch that the library
        cannot see must be declared with `bias_written()`."""
        native.check(self._lib.bpr_set_bias_tracking(self._ctx, int(bool(on))))

    def bias_written(self) -> None:
        native.check(self._lib.bpr_bias_written(self._ctx))

    def train_strict(self, users, pos, batch_size: int, sampler: int = NEG_UNIFORM,
                     neg: Optional[torch.Tensor] = None, adaptive_p: float = 0.01, seed: int = 0,
                     offset: int = 0, refresh_every: int = 0,
                     scalars: Optional[torch.Tensor] = None) -> None:
        """STRICT mini-batch epoch (any optimizer) with the batch loop inside the library."""
        self._sync_stream()
        if users.dtype != torch.int32 or pos.dtype != torch.int32:
            raise ValueError("train_strict takes int32 id tensors")
        if neg is None:
            if sampler == NEG_GIVEN:
                raise ValueError("sampler NEG_GIVEN needs `neg`")
            neg = torch.empty(max(batch_size, 1), dtype=torch.int32, device=self.device)
        native.check(self._lib.bpr_train_strict(self._ctx, users.data_ptr(), pos.data_ptr(),
                                                neg.data_ptr(), users.numel(), batch_size, sampler,
                                                adaptive_p, seed, offset, refresh_every,
                                                _ptr(scalars)))

    def train_stream_batched(self, users, pos, batch_size: int, sampler: int = NEG_UNIFORM,
                             neg: Optional[torch.Tensor] = None, adaptive_p: float = 0.01,
                             seed: int = 0, offset: int = 0, max_inflight: int = 0,
                             scalars: Optional[torch.Tensor] = None) -> None:
        """BATCHED STREAM over users/pos (int32, on device, shuffled — not grouped by user) in one
        launch, any optimizer: virtual mini-batches of `batch_size` consecutive triples, one
        torch.optim step per row and batch (see include/bprcore.h)."""
        self._sync_stream()
        if users.dtype != torch.int32 or pos.dtype != torch.int32:
            raise ValueError("train_stream_batched takes int32 id tensors")
        if sampler == NEG_GIVEN and neg is None:
            raise ValueError("sampler NEG_GIVEN needs `neg`")
        native.check(self._lib.bpr_train_stream_batched(
            self._ctx, users.data_ptr(), pos.data_ptr(), _ptr(neg), users.numel(), batch_size,
            sampler, adaptive_p, seed, offset, max_inflight, _ptr(scalars)))

    def shuffle_epoch(self, users: torch.Tensor, pos: torch.Tensor, seed: int,
                      out: Optional[tuple[torch.Tensor, torch.Tensor]] = None):
        """Seeded pseudo-random permutation of the training triples (on device)."""
        self._sync_stream()
        if users.dtype != torch.int32 or pos.dtype != torch.int32:
            raise ValueError("shuffle_epoch takes int32 id tensors")
        uo, po = out if out is not None else (torch.empty_like(users), torch.empty_like(pos))
        native.check(self._lib.bpr_shuffle_epoch(self._ctx, users.data_ptr(), pos.data_ptr(),
                                                 users.numel(), seed, uo.data_ptr(), po.data_ptr()))
        return uo, po

    def set_stream_opts(self, grouped_by_user: bool, run_len: int = 0) -> None:
        native.check(self._lib.bpr_set_stream_opts(self._ctx, int(grouped_by_user), run_len))

    def stream_run_len(self) -> int:
        """Run length the last STREAM launch used (run_len = 0 lets the library pick it)."""
        return int(self._lib.bpr_stream_run_len(self._ctx))

    def set_hot_lds(self, rows: int, always: bool = False) -> None:
        """``bpr_set_hot_lds``: the `rows` most popular hot rows take a CU's updates in LDS (flushed at the
        workgroup's exit): a hot row is one launch stale across CUs — switch it on by ``fast.lag_within_budget``.
        ``always``: also for launches that do not fill the chip (tests)."""
        native.check(self._lib.bpr_set_hot_lds(self._ctx, int(rows), int(always)))

    def stream_lds_rows(self) -> int:
        """LDS rows of the last STREAM launch (0: the plain kernel ran)."""
        return int(self._lib.bpr_stream_lds_rows(self._ctx))

    def hot_fold(self) -> None:
        """Fold the hot rows' deltas an asynchronous cut left in the block (no-op otherwise): after it
        the item table's storage is whole for a direct reader (eval, checkpoint)."""
        self._sync_stream()
        native.check(self._lib.bpr_hot_fold(self._ctx))

    def set_hot_rows(self, hot_rows: int = 256, replicas: int = 1) -> None:
        """Replica delta rows for the most popular item rows in STREAM mode (0 = off); takes
        effect at the next plan_epoch."""
        native.check(self._lib.bpr_set_hot_rows(self._ctx, hot_rows, replicas))

    # ---- two-tier reconciliation, hot tier (revisit_bpr.distributed.ItemSync drives it)
    def set_hot_items(self, items: torch.Tensor, counts: Optional[torch.Tensor] = None) -> None:
        """The hot set as given (the ranks of a job agree on it): `items` int32 ids in the canonical
        order of the exchange buffers; `counts` [I] positives per item (steers the placement)."""
        import numpy as np

        it = np.ascontiguousarray(items.detach().cpu().numpy(), dtype=np.int32)
        cn = None if counts is None else np.ascontiguousarray(counts.detach().cpu().numpy(), dtype=np.uint32)
        native.check(self._lib.bpr_set_hot_items(
            self._ctx, it.ctypes.data_as(ctypes.c_void_p), int(it.size),
            None if cn is None else cn.ctypes.data_as(ctypes.c_void_p)))

    def hot_rows(self) -> int:
        v = ctypes.c_int32()
        native.check(self._lib.bpr_hot_rows(self._ctx, ctypes.byref(v)))
        return v.value

    def hot_tier_begin(self, hot_base: torch.Tensor) -> None:
        self._sync_stream()
        native.check(self._lib.bpr_hot_tier_begin(self._ctx, hot_base.data_ptr()))

    def hot_exchange(self, hot_base: torch.Tensor, tot: torch.Tensor, fold_prev: bool, cut: bool,
                     cold_base: Optional[torch.Tensor] = None) -> None:
        self._sync_stream()
        native.check(self._lib.bpr_hot_exchange(self._ctx, hot_base.data_ptr(), tot.data_ptr(),
                                                int(fold_prev), int(cut), _ptr(cold_base)))

    def sync_cut(self, hot_base, hot_tot, hot_fold_prev: bool, cold_base, cold_own, cold_tot, scale: float,
                 cold_mode: int) -> None:
        """`bpr_sync_cut`: hot-tier step + cold-tier step + the cut of the next snapshot's keys in one
        pass over the item table (after a `train_stream(cut=True)` under the hot tier)."""
        self._sync_stream()
        native.check(self._lib.bpr_sync_cut(self._ctx, _ptr(hot_base), _ptr(hot_tot), int(hot_fold_prev),
                                            _ptr(cold_base), _ptr(cold_own), _ptr(cold_tot), scale, cold_mode))

    def hot_tier_end(self) -> None:
        native.check(self._lib.bpr_hot_tier_end(self._ctx))

    def plan_epoch(self, users: torch.Tensor, pos: torch.Tensor, chunk: int, seed: int,
                   out: Optional[tuple[torch.Tensor, torch.Tensor]] = None, sorted_input: bool = False):
        """Shuffle the training triples into chunks of `chunk`, each grouped by user (on device).
        sorted_input: a PROMISE that `users` is sorted by user id (`users_sorted` checks it once per training
        set): the same plan in one radix pass instead of three (bpr_set_tuning "plan_input_sorted")."""
        self._sync_stream()
        if users.dtype != torch.int32 or pos.dtype != torch.int32:
            raise ValueError("plan_epoch takes int32 id tensors")
        if bool(sorted_input) != getattr(self, "_plan_sorted", False):
            self._plan_sorted = bool(sorted_input)
            self.set_tuning("plan_input_sorted", int(self._plan_sorted))
        uo, po = out if out is not None else (torch.empty_like(users), torch.empty_like(pos))
        native.check(self._lib.bpr_plan_epoch(self._ctx, users.data_ptr(), pos.data_ptr(),
                                              users.numel(), chunk, seed, uo.data_ptr(),
                                              po.data_ptr()))
        return uo, po

    @staticmethod
    def users_sorted(users: torch.Tensor) -> bool:
        """Is the triple list in CSR order (sorted by user)?  One pass + one host read: call it once per training set."""
        return bool(users.numel() < 2 or (users[1:] >= users[:-1]).all().item())

    def plan_chunk(self, users: torch.Tensor, pos: torch.Tensor, chunk: int, seed: int, index: int,
                   out: tuple[torch.Tensor, torch.Tensor], on_side: bool = False) -> None:
        """Chunk `index` of the plan `plan_epoch(seed)` would make (same members, grouped by user),
        alone; on_side: queued on the split refresh's side stream behind the sort in flight."""
        if not on_side:
            self._sync_stream()
        native.check(self._lib.bpr_plan_chunk(self._ctx, users.data_ptr(), pos.data_ptr(), users.numel(),
                                              chunk, seed, index, out[0].data_ptr(), out[1].data_ptr(),
                                              int(on_side)))

    def flush_lazy(self) -> None:
        self._sync_stream()
        native.check(self._lib.bpr_flush_lazy(self._ctx))

    def flush_items(self) -> None:
        """flush_lazy for the item table (+ bias) only (before an item reconciliation)."""
        self._sync_stream()
        native.check(self._lib.bpr_flush_items(self._ctx))

    @property
    def step_count(self) -> int:
        v = ctypes.c_int64()
        native.check(self._lib.bpr_get_step_host(self._ctx, ctypes.byref(v)))
        return v.value

    def set_step(self, step: int) -> None:
        native.check(self._lib.bpr_set_step(self._ctx, step))

    def set_sampler_iter(self, iteration: int) -> None:
        """Batch counter of the train_strict loop (AdaptiveSampler._iteration_cnt)."""
        native.check(self._lib.bpr_set_sampler_iter(self._ctx, iteration))

    # ---- measurement ------------------------------------------------------------------------
    def timing_enable(self, on=True) -> None:
        """hipEvent timing of the dominant kernel: True / 1 = every launch, N > 1 = every N-th launch
        (the event records idle the stream for a few microseconds), False / 0 = off."""
        native.check(self._lib.bpr_timing_enable(self._ctx, int(on)))

    def timing_read(self) -> tuple[float, int]:
        ms, n = ctypes.c_double(), ctypes.c_int64()
        native.check(self._lib.bpr_timing_read_host(self._ctx, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value
