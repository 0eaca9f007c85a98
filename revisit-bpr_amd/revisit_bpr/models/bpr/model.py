"""BPR model + MF scorer behind the reference's API (revisit_bpr/models/bpr/model.py:8-153).

Training forward on a ROCm device does not build an autograd graph: it calls the HIP engine
(STRICT phase A: logits, loss terms and per-row gradients in one kernel) and returns a `loss`
tensor whose `.backward()` merely arms the pending update.  The stock ``torch.optim`` optimizer
built over ``model.parameters()`` is intercepted by a global step pre-hook that reads its
hyper-parameters, binds its state tensors and launches STRICT phase B on the touched rows; the
optimizer's own ``step()`` then finds ``grad is None`` everywhere and does nothing.  So the
reference loop

    batch["neg"] = sampler.sample(batch); out = model(batch)
    accelerator.backward(out["loss"]); optimizer.step(); optimizer.zero_grad()

(example.py:176-180, experiments/trainer.py:67-82) runs unchanged with identical results.

``set_backend("torch")`` switches to a plain PyTorch-ROCm restatement (dense autograd, as the
reference computes it) — the comparison leg of BASELINE config 2, never a silent fallback: with the
default backend "hip", training on a non-ROCm tensor raises.
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch
from torch.nn import init

from revisit_bpr.models.bpr.loss import Loss

_BACKEND = "hip"


def set_backend(name: str) -> None:
    """"hip" (default): fused HIP engine, fails loudly without a GPU.  "torch": dense autograd."""
    global _BACKEND
    if name not in ("hip", "torch"):
        raise ValueError(f"unknown backend {name!r}")
    _BACKEND = name


def get_backend() -> str:
    return _BACKEND


class BaseLogitModel(torch.nn.Module):
    def get_features(self) -> dict[str, torch.Tensor]:
        return {}


class MF(BaseLogitModel):
    """x_ui = <p_u, q_i> (+ b_i) (+ b_u).  Reference: model.py:96-153."""

    def __init__(self, user_emb: torch.nn.Embedding, item_emb: torch.nn.Embedding,
                 item_bias: bool = False, user_bias: bool = False) -> None:
        super().__init__()
        self._user_emb = user_emb
        self._item_emb = item_emb
        for name, table, wanted in (("_item_bias", item_emb, item_bias),
                                    ("_user_bias", user_emb, user_bias)):
            if wanted:
                setattr(self, name, torch.nn.Parameter(torch.empty(table.num_embeddings)))
            else:
                self.register_parameter(name, None)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        # same draw order as the reference (model.py:117-129): users, then items; U[0,1) - 0.5, / d
        with torch.no_grad():
            for emb in (self._user_emb, self._item_emb):
                emb.weight.uniform_().add_(-0.5).div_(emb.embedding_dim)
                if emb.padding_idx is not None:
                    emb.weight[emb.padding_idx].fill_(0)
        for bias in (self._item_bias, self._user_bias):
            if bias is not None:
                init.zeros_(bias)

    def forward(self, user: torch.Tensor, item: torch.Tensor, _: dict) -> torch.Tensor:
        # user [B], item [B, ...] -> logits [B, ...]
        p = self._user_emb(user)
        q = self._item_emb(item)
        logits = torch.einsum("bh,b...h->b...", p, q)
        if self._item_bias is not None:
            logits = logits + self._item_bias[item]
        if self._user_bias is not None:
            ub = self._user_bias[user]
            logits = logits + ub.reshape(ub.shape + (1,) * (logits.dim() - ub.dim()))
        return logits

    def get_features(self) -> dict[str, torch.Tensor]:
        return {"user": self._user_emb.weight, "item": self._item_emb.weight,
                "user_bias": self._user_bias, "item_bias": self._item_bias}


class _ArmUpdate(torch.autograd.Function):
    """loss.backward() for the fused path: no gradient flows, the pending update is armed."""

    @staticmethod
    def forward(ctx, anchor: torch.Tensor, value: torch.Tensor, model_ref):
        ctx.model_ref = model_ref
        return value.clone()

    @staticmethod
    def backward(ctx, grad_out):
        model = ctx.model_ref()
        if model is not None:
            model._armed = True
        return None, None, None


# parameters of fused models: id(param) -> weakref(Model); consulted by the optimizer hook
_FUSED_PARAMS: "dict[int, weakref.ReferenceType]" = {}
_HOOK_INSTALLED = False

_OPT_KINDS = {"SGD": 0, "Adam": 2, "RMSprop": 3}


def _optimizer_pre_hook(optimizer, args, kwargs):
    models = {}
    for group in optimizer.param_groups:
        for prm in group["params"]:
            ref = _FUSED_PARAMS.get(id(prm))
            model = ref() if ref is not None else None
            if model is not None and model._engine is not None:
                seen = models.setdefault(id(model), (model, group, []))
                if seen[1] is not group:
                    raise NotImplementedError(
                        "the fused BPR update applies ONE set of hyper-parameters to the user, item "
                        "and bias tables: keep them in a single param group (or use "
                        "set_backend('torch'))")
                seen[2].append(prm)
    for model, group, prms in models.values():
        if model._pending and len(prms) != len(model._fused_params()):
            raise NotImplementedError(
                "the fused BPR tables must all belong to the optimizer that steps them")
        model._apply_pending(optimizer, group)
    return None


def _install_hook() -> None:
    global _HOOK_INSTALLED
    if not _HOOK_INSTALLED:
        from torch.optim.optimizer import register_optimizer_step_pre_hook

        register_optimizer_step_pre_hook(_optimizer_pre_hook)
        _HOOK_INSTALLED = True


class _SeenItemScorer(BaseLogitModel):
    """Shared frame of the two item-to-item scorers (reference model.py:156-255; no reference config
    instantiates them — they are kept for the import surface, as plain PyTorch modules: the HIP engine
    is the MF path).  A candidate's logit is the sum, over the user's seen items, of an item-item
    similarity; a seen item that is itself among the row's candidates contributes nothing."""

    def __init__(self, num_items: int, width: int, padding_idx: int, bias: bool) -> None:
        super().__init__()
        self._padding_idx = padding_idx
        self._weights = torch.nn.Parameter(torch.empty(num_items, width))
        if bias:
            self._bias = torch.nn.Parameter(torch.empty(num_items))
        else:
            self.register_parameter("_bias", None)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        with torch.no_grad():
            self._weights.uniform_()
            self._weights[self._padding_idx].zero_()
            if self._bias is not None:
                self._bias.zero_()

    @staticmethod
    def _kept(item: torch.Tensor, seen: torch.Tensor) -> torch.Tensor:
        """[batch, seen] 1.0 where the seen item is NOT one of the row's candidates"""
        hit = (seen.unsqueeze(1) == item.unsqueeze(2)).any(dim=1)
        return (~hit).to(torch.float32)

    def get_features(self) -> dict[str, torch.Tensor]:
        return {"item": self._weights, "bias": self._bias}


class ItemKNN(_SeenItemScorer):
    """similarity(i, s) = <w_i, w_s> with w in R^hidden_dim (reference model.py:156-199)."""

    def __init__(self, num_items: int, hidden_dim: int, padding_idx: int = 0, bias: bool = False) -> None:
        super().__init__(num_items, hidden_dim, padding_idx, bias)

    def forward(self, _user, item: torch.Tensor, other: dict) -> torch.Tensor:
        seen = other["seen_items"]
        profile = (self._weights[seen] * self._kept(item, seen).unsqueeze(-1)).sum(dim=1)  # [batch, hidden]
        logits = torch.bmm(self._weights[item], profile.unsqueeze(-1)).squeeze(-1)
        return logits if self._bias is None else logits + self._bias[item]


class FreeItemKNN(_SeenItemScorer):
    """similarity(i, s) = W[i, s], a free num_items x num_items matrix (reference model.py:201-255)."""

    def __init__(self, num_items: int, padding_idx: int = 0, bias: bool = False) -> None:
        super().__init__(num_items, num_items, padding_idx, bias)

    def reset_parameters(self) -> None:
        super().reset_parameters()

    def forward(self, _user, item: torch.Tensor, other: Optional[dict]) -> torch.Tensor:
        if not other or "seen_items" not in other:
            raise ValueError("seen_items should be present")
        seen = other["seen_items"]
        rows = self._weights[item]                                         # [batch, items, num_items]
        sims = rows.gather(-1, seen.unsqueeze(1).expand(-1, item.size(-1), -1))
        logits = (sims * self._kept(item, seen).unsqueeze(1)).sum(dim=-1)
        return logits if self._bias is None else logits + self._bias[item]


class Model(torch.nn.Module):
    """The BPR model (reference: model.py:13-93).

    logits_model : scorer producing logits for user-item pairs (MF for the fused path)
    reg_alphas   : {"user","item","neg","all"}; `all` overrides, missing neg -> item
    fuse_forward : accepted for API compatibility (the fused kernel always scores pos and neg together)
    """

    def __init__(self, logits_model: BaseLogitModel, reg_alphas: Optional[dict] = None,
                 fuse_forward: bool = False) -> None:
        super().__init__()
        self.logits_model = logits_model
        self._reg_alphas = reg_alphas or {}
        self._fuse_forward = fuse_forward
        self._loss = Loss(size_average=False)
        self._engine = None
        self._engine_key = None
        self._armed = False
        self._pending = False
        self._opt_sig = None
        self._state_sig = None
        self._reg_n = 1
        self._anchor = None

    # ---- fused engine plumbing ------------------------------------------------------------
    def _fusable(self) -> bool:
        lm = self.logits_model
        return isinstance(lm, MF) and lm._user_emb.weight.dtype == torch.float32

    def engine(self):
        """The HIP engine bound to the current parameter storage (created on first use)."""
        from revisit_bpr.engine import Engine, resolve_reg_alphas

        lm = self.logits_model
        P, Q, b = lm._user_emb.weight, lm._item_emb.weight, lm._item_bias
        key = (P.data_ptr(), Q.data_ptr(), None if b is None else b.data_ptr(), P.device)
        if self._engine is None or key != self._engine_key:
            if not P.is_cuda:
                raise RuntimeError(
                    "revisit_bpr BPR training runs on the HIP engine and needs the model on a ROCm "
                    "device (model.to('cuda')); there is no CPU fallback. For the dense PyTorch "
                    "restatement call revisit_bpr.models.bpr.set_backend('torch') explicitly.")
            self._engine = Engine(P.data, Q.data, None if b is None else b.data,
                                  pad_user=lm._user_emb.padding_idx,
                                  pad_item=lm._item_emb.padding_idx)
            self._engine.set_reg(*resolve_reg_alphas(self._reg_alphas))
            self._engine_key = key
            self._opt_sig = None
            self._state_sig = None
            self._reg_n = 1
            self._pending = self._armed = False
            for prm in (P, Q, b):
                if prm is not None:
                    _FUSED_PARAMS[id(prm)] = weakref.ref(self)
            _install_hook()
        return self._engine

    def bind_seen_csr(self, indptr: torch.Tensor, indices: torch.Tensor) -> None:
        """Seen-items CSR over users (int64 [U+1], int32 [nnz] sorted per row) for on-device sampling."""
        self.engine().bind_seen_csr(indptr, indices)
        self._has_csr = True

    def _apply_pending(self, optimizer, group) -> None:
        """optimizer.step() for the fused path (called from the global step pre-hook)."""
        if not self._pending:
            return
        eng = self._engine
        if not self._armed:  # forward without backward: torch would have no gradient either
            eng.discard_grad()
            self._pending = False
            return
        kind = self._configure_optimizer(optimizer, group)
        eng.apply()
        if kind != 0:
            for prm in self._fused_params():
                st = optimizer.state[prm]
                st["step"] = st.get("step", 0) + 1 if not torch.is_tensor(st.get("step")) \
                    else st["step"] + 1
        self._pending = self._armed = False

    def _configure_optimizer(self, optimizer, group) -> int:
        """Mirror a torch.optim optimizer (kind + hyper-parameters of `group`, read on every call so
        LR changes are honoured) into the engine and bind its state tensors."""
        eng = self._engine
        name = type(optimizer).__name__
        if name not in _OPT_KINDS:
            raise NotImplementedError(
                f"fused BPR supports torch.optim SGD / Adam / RMSprop, got {name}; use "
                "set_backend('torch') for other optimizers")
        if group.get("weight_decay", 0) or group.get("maximize", False) or group.get("amsgrad", False) \
                or group.get("centered", False):
            raise NotImplementedError("weight_decay / maximize / amsgrad / centered are not fused")
        kind = _OPT_KINDS[name]
        if name == "SGD" and group.get("momentum", 0) != 0:
            kind = 1
        sig = (kind, float(group["lr"]), group.get("momentum", 0.0), group.get("dampening", 0.0),
               bool(group.get("nesterov", False)), tuple(group.get("betas", (0.9, 0.999))),
               group.get("eps", 1e-8), group.get("alpha", 0.99))
        if sig != self._opt_sig:
            # rows are brought to "now" lazily with the CURRENT hyper-parameters: a changed lr /
            # betas / momentum (LR scheduler, manual edit) must not be applied to the zero-gradient
            # steps a row missed under the old ones — replay those first
            if self._opt_sig is not None and self._opt_sig[0] != 0 and eng.step_count > 0:
                eng.flush_lazy()
            eng.set_optimizer(kind, lr=sig[1], momentum=sig[2], dampening=sig[3], nesterov=sig[4],
                              betas=sig[5], eps=sig[6], alpha=sig[7])
            self._opt_sig = sig
            self._state_sig = None
        # the state tensors are looked up on every call: optimizer.load_state_dict() (checkpoint
        # resume) replaces them, and the engine must follow
        self._bind_state(optimizer, kind)
        return kind

    def _reset_reg(self) -> None:
        """One item per triple again: a forward() with n items per row left alpha_user / n in the
        engine (it would silently regularise the epoch drivers' user rows with it)."""
        if self._reg_n != 1:
            from revisit_bpr.engine import resolve_reg_alphas

            self.engine().set_reg(*resolve_reg_alphas(self._reg_alphas))
            self._reg_n = 1

    def train_strict(self, optimizer, users: torch.Tensor, items: torch.Tensor, batch_size: int,
                     sampler: int, adaptive_p: float = 0.0, seed: int = 0, offset: int = 0,
                     refresh_every: int = 0, scalars: Optional[torch.Tensor] = None) -> int:
        """The reference's inner loop — for batch: sample negatives, model(batch), backward,
        optimizer.step() — for ALL batches of `users` / `items` (int32, device) inside the library
        (`bpr_train_strict`): the same mini-batch semantics as the per-batch API, without the
        Python round trip per batch.  Returns the number of optimizer steps taken."""
        eng = self.engine()
        if self._pending:
            raise RuntimeError("a forward() is waiting for optimizer.step()")
        self._reset_reg()
        groups = [g for g in optimizer.param_groups
                  if any(id(p) in _FUSED_PARAMS for p in g["params"])]
        if len(groups) != 1:
            raise ValueError("the fused tables must sit in exactly one param group")
        kind = self._configure_optimizer(optimizer, groups[0])
        eng.train_strict(users, items, batch_size, sampler=sampler, adaptive_p=adaptive_p, seed=seed,
                         offset=offset, refresh_every=refresh_every, scalars=scalars)
        steps = (users.numel() + batch_size - 1) // batch_size
        if kind != 0:
            for prm in self._fused_params():
                st = optimizer.state[prm]
                st["step"] = st.get("step", 0) + steps
        return steps

    def train_stream_batched(self, optimizer, users: torch.Tensor, items: torch.Tensor,
                             batch_size: int, sampler: int, adaptive_p: float = 0.0, seed: int = 0,
                             offset: int = 0, max_inflight: int = 0,
                             scalars: Optional[torch.Tensor] = None) -> int:
        """The same loop as train_strict — virtual mini-batches of `batch_size` consecutive
        triples, one torch.optim step per batch — as ONE fused launch (`bpr_train_stream_batched`,
        any optimizer): triples of neighbouring batches run concurrently, so a gradient may see
        rows that are a few steps stale (bounded by `max_inflight`).  Returns the steps taken."""
        eng = self.engine()
        self._reset_reg()
        if self._pending:
            raise RuntimeError("a forward() is waiting for optimizer.step()")
        groups = [g for g in optimizer.param_groups
                  if any(id(p) in _FUSED_PARAMS for p in g["params"])]
        if len(groups) != 1:
            raise ValueError("the fused tables must sit in exactly one param group")
        kind = self._configure_optimizer(optimizer, groups[0])
        eng.train_stream_batched(users, items, batch_size, sampler=sampler, adaptive_p=adaptive_p,
                                 seed=seed, offset=offset, max_inflight=max_inflight,
                                 scalars=scalars)
        steps = (users.numel() + batch_size - 1) // batch_size
        if kind != 0:
            for prm in self._fused_params():
                st = optimizer.state[prm]
                st["step"] = st.get("step", 0) + steps
        return steps

    def _fused_params(self):
        lm = self.logits_model
        return [t for t in (lm._user_emb.weight, lm._item_emb.weight, lm._item_bias) if t is not None]

    def _bind_state(self, optimizer, kind: int) -> None:
        """Create torch-compatible optimizer state tensors and hand their storage to the engine.
        Re-binds whenever the tensors in ``optimizer.state`` are not the bound ones any more
        (``load_state_dict``), and then also restores the engine's step counter from
        ``state["step"]`` — the loaded rows are "as of that step" (checkpoints are written after
        ``Model.sync()``), so bias corrections and the lazy replay continue where they stopped."""
        if kind == 0:
            return
        rms_mom = kind == 3 and self._opt_sig is not None and self._opt_sig[2] != 0
        keys = {1: ("momentum_buffer", None), 2: ("exp_avg", "exp_avg_sq"),
                3: ("momentum_buffer" if rms_mom else None, "square_avg")}[kind]
        bufs, created = [], False
        for prm in self._fused_params():
            st = optimizer.state[prm]
            pair = []
            for k in keys:
                if k is None:
                    pair.append(None)
                    continue
                if k not in st or st[k] is None:
                    st[k] = torch.zeros_like(prm, memory_format=torch.preserve_format)
                    created = True
                pair.append(st[k])
            st.setdefault("step", 0)
            bufs.append(pair)
        sig = tuple(None if t is None else t.data_ptr() for pair in bufs for t in pair)
        if sig == self._state_sig:
            return
        for pair in bufs:
            for t in pair:
                if t is not None and (not t.is_contiguous() or t.dtype != torch.float32):
                    raise ValueError("optimizer state tensors must be contiguous float32")
        (mP, vP), (mQ, vQ) = bufs[0], bufs[1]
        mb, vb = bufs[2] if len(bufs) > 2 else (None, None)
        eng = self._engine
        if self._state_sig is not None or not created:
            # foreign tensors (a loaded checkpoint): rows in them are current as of `step`
            step = max(int(optimizer.state[prm]["step"]) for prm in self._fused_params())
            if kind == 1 and step == 0 and not created:
                step = 1  # torch's SGD keeps no step: a present momentum_buffer means "not the first"
            # always: set_step also resets the per-row marks (lastP / lastQ, the batched stream's
            # headers) — a checkpoint with the SAME step count loaded while rows were lazily behind
            # (an in-process restore-best) must not replay their missed steps on the loaded state
            eng.set_step(step)
        eng.bind_opt_state(mP, vP, mQ, vQ, mb, vb)
        self._state_sig = sig

    def sync(self) -> None:
        """Bring every row to the current optimizer step (lazy dense-optimizer replay): called
        automatically before eval and state_dict."""
        if self._engine is None:
            return
        if self._pending:
            if self._armed:
                raise RuntimeError("loss.backward() was called but optimizer.step() was not: step "
                                   "(or zero the gradients by a new forward) before eval / state_dict")
            # forward without backward: torch would hold no gradient either
            self._engine.discard_grad()
            self._pending = False
        self._engine.flush_lazy()

    def train(self, mode: bool = True):
        if not mode:
            self.sync()
        return super().train(mode)

    def state_dict(self, *args, **kwargs):
        self.sync()
        return super().state_dict(*args, **kwargs)

    # ---- forward ----------------------------------------------------------------------------
    def forward(self, inputs: dict[str, torch.Tensor]) -> dict[str, torch.Tensor]:
        # inputs.user [B]; inputs.item, inputs.neg [B, n]
        if not self.training:
            logits = self.logits_model(inputs["user"], inputs["item"], inputs)
            mask = inputs.get("mask")
            if mask is not None:
                logits.masked_fill_(mask.eq(0), -1e13)
            return {"logits": logits}
        if _BACKEND == "hip" and self._fusable():
            return self._forward_fused(inputs)
        if _BACKEND == "hip" and not inputs["user"].is_cuda:
            raise RuntimeError("BPR training needs a ROCm device (no CPU fallback); "
                               "set_backend('torch') selects the dense PyTorch restatement")
        return self._forward_torch(inputs)

    def _forward_fused(self, inputs):
        eng = self.engine()
        if self._pending and not self._armed:
            eng.discard_grad()  # previous forward was never backpropagated
            self._pending = False
        user, item, neg = inputs["user"], inputs["item"], inputs["neg"]
        shape = item.shape
        n = item.numel() // max(user.numel(), 1)
        users = user if n == 1 else user.repeat_interleave(n)
        if n != self._reg_n:
            # Model.regularization counts the user term once per ROW and the item terms once per
            # (row, item) pair (model.py:87-93): with n items per row the n triples of a row share
            # one user term
            from revisit_bpr.engine import resolve_reg_alphas

            a_user, a_item, a_neg = resolve_reg_alphas(self._reg_alphas)
            eng.set_reg(a_user / n, a_item, a_neg)
            self._reg_n = n
        lp, ln, sc = eng.forward_grad(users, item.reshape(-1), neg.reshape(-1))
        self._pending = True
        self._armed = False
        if self._anchor is None or self._anchor.device != lp.device:
            self._anchor = torch.zeros((), device=lp.device, requires_grad=True)
        lp, ln = lp.view(shape), ln.view(shape)
        ub = self.logits_model._user_bias
        if ub is not None:
            # a user bias shifts both logits equally: it cancels in x = pos - neg, so its gradient
            # is exactly zero and only the reported logits carry it
            shift = ub.detach()[user].reshape(user.shape + (1,) * (lp.dim() - user.dim()))
            lp, ln = lp + shift, ln + shift
        out = {"logits_pos": lp, "logits_neg": ln, "logits": lp - ln,
               "bpr_loss": sc[0], "l2_reg": sc[1]}
        out["loss"] = _ArmUpdate.apply(self._anchor, sc[0] + sc[1], weakref.ref(self))
        return out

    def _forward_torch(self, inputs):
        n = inputs["item"].size(-1)
        both = self.logits_model(inputs["user"], torch.hstack((inputs["item"], inputs["neg"])), inputs)
        out = {"logits_pos": both[:, :n], "logits_neg": both[:, n:]}
        out["logits"] = out["logits_pos"] - out["logits_neg"]
        out["bpr_loss"] = self._loss(out["logits"]).sum()
        out["l2_reg"] = self.regularization(inputs).sum()
        out["loss"] = out["bpr_loss"] + out["l2_reg"]
        return out

    def regularization(self, inputs: dict[str, torch.Tensor]) -> torch.Tensor:
        """0.5 (a_item |q_i|^2 + a_neg |q_j|^2 + a_user |p_u|^2) per row (reference: model.py:70-93)."""
        from revisit_bpr.engine import resolve_reg_alphas

        feats = self.logits_model.get_features()
        if not feats or all(self._reg_alphas.get(k) is None for k in ("all", "user", "item", "neg")):
            return torch.tensor(0)
        a_user, a_item, a_neg = resolve_reg_alphas(self._reg_alphas)
        sq = lambda t: t.pow(2).flatten(1).sum(1)  # noqa: E731
        term = a_item * sq(feats["item"][inputs["item"]]) + a_neg * sq(feats["item"][inputs["neg"]])
        if feats.get("user") is not None:
            term = term + a_user * sq(feats["user"][inputs["user"]])
        return term.mean() / 2 if self._loss.size_average else term / 2
