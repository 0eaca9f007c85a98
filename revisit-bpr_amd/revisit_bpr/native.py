"""ctypes binding of libbprcore.so (include/bprcore.h) — the only door into the HIP engine.

There is no CPU fallback: if the shared library is missing or no MI355X is visible, every entry
point raises.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C revisit-bpr_amd/csrc``.
"""
from __future__ import annotations

import ctypes
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p
from pathlib import Path

import os

# BPR_LIB_PATH: tuning builds only (A/B of compile-time knobs on the GPU box)
LIB_PATH = Path(os.environ.get("BPR_LIB_PATH") or
                Path(__file__).resolve().parent.parent / "libbprcore.so")

OK = 0
OPT_SGD, OPT_MOMENTUM, OPT_ADAM, OPT_RMSPROP = 0, 1, 2, 3
MODE_STRICT, MODE_STREAM = 0, 1
NEG_GIVEN, NEG_UNIFORM, NEG_ADAPTIVE = 0, 1, 2
SCALARS = 4


class OptParams(ctypes.Structure):
    """struct bpr_opt_params"""

    _fields_ = [
        ("lr", c_float),
        ("momentum", c_float),
        ("dampening", c_float),
        ("nesterov", c_int32),
        ("beta1", c_float),
        ("beta2", c_float),
        ("eps", c_float),
        ("alpha", c_float),
    ]


class BprError(RuntimeError):
    def __init__(self, code: int, msg: str) -> None:
        super().__init__(f"libbprcore error {code}: {msg}")
        self.code = code


# name -> (restype, argtypes); mirrors include/bprcore.h one to one
SIGNATURES = {
    "bpr_version": (c_int, []),
    "bpr_last_error": (c_char_p, []),
    "bpr_ctx_create": (c_int, [POINTER(c_void_p), c_int, c_void_p]),
    "bpr_ctx_destroy": (c_int, [c_void_p]),
    "bpr_set_stream": (c_int, [c_void_p, c_void_p]),
    "bpr_bind_tables": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p,
                                c_int32, c_int32]),
    "bpr_bind_seen_csr": (c_int, [c_void_p, c_void_p, c_void_p]),
    "bpr_bind_item_weights": (c_int, [c_void_p, c_void_p, c_void_p]),
    "bpr_set_reg": (c_int, [c_void_p, c_float, c_float, c_float]),
    "bpr_set_tuning": (c_int, [c_void_p, c_char_p, c_int32]),
    "bpr_adaptive_snapshot_partial": (c_int, [c_void_p, POINTER(c_int32)]),
    "bpr_set_bias_tracking": (c_int, [c_void_p, c_int32]),
    "bpr_bias_written": (c_int, [c_void_p]),
    "bpr_set_optimizer": (c_int, [c_void_p, c_int32, POINTER(OptParams)]),
    "bpr_bind_opt_state": (c_int, [c_void_p] + [c_void_p] * 6),
    "bpr_sample_uniform": (c_int, [c_void_p, c_void_p, c_int64, c_uint64, c_uint64, c_void_p]),
    "bpr_adaptive_refresh": (c_int, [c_void_p]),
    "bpr_adaptive_refresh_begin": (c_int, [c_void_p]),
    "bpr_adaptive_refresh_commit": (c_int, [c_void_p]),
    "bpr_adaptive_refresh_pending": (c_int, [c_void_p, POINTER(c_int32)]),
    "bpr_adaptive_refresh_part": (c_int, [c_void_p, c_int32, c_int32]),
    "bpr_adaptive_refresh_publish": (c_int, [c_void_p]),
    "bpr_adaptive_snapshot_ptrs": (c_int, [c_void_p, c_int32, POINTER(c_void_p), POINTER(c_void_p)]),
    "bpr_comm_unique_id": (c_int, [c_void_p]),
    "bpr_comm_init": (c_int, [c_void_p, c_void_p, c_int32, c_int32]),
    "bpr_comm_destroy": (c_int, [c_void_p]),
    "bpr_item_sync": (c_int, [c_void_p]),
    "bpr_item_sync_finish": (c_int, [c_void_p]),
    "bpr_item_sync_rebase": (c_int, [c_void_p]),
    "bpr_comm_hot_tier": (c_int, [c_void_p, c_void_p, c_int32, c_void_p]),
    "bpr_hot_sync": (c_int, [c_void_p]),
    "bpr_set_side_stream": (c_int, [c_void_p, c_void_p]),
    "bpr_stream_create": (c_int, [c_int, c_void_p, c_int32, POINTER(c_void_p)]),
    "bpr_stream_destroy": (c_int, [c_void_p]),
    "bpr_sample_adaptive": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_uint64, c_uint64,
                                    c_void_p, c_void_p, c_void_p]),
    "bpr_adaptive_pick": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "bpr_adaptive_get_snapshot": (c_int, [c_void_p, c_void_p, c_void_p]),
    "bpr_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                            c_void_p]),
    "bpr_forward_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                 c_void_p, c_void_p]),
    "bpr_apply": (c_int, [c_void_p]),
    "bpr_discard_grad": (c_int, [c_void_p]),
    "bpr_get_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "bpr_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_float,
                         c_uint64, c_uint64, c_void_p, c_void_p, c_void_p]),
    "bpr_train_stream": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float,
                                 c_uint64, c_uint64, c_int64, c_void_p]),
    "bpr_train_stream_cut": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float,
                                     c_uint64, c_uint64, c_int64, c_void_p]),
    "bpr_train_stream_acut": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float,
                                      c_uint64, c_uint64, c_int64, c_void_p]),
    "bpr_hot_fold": (c_int, [c_void_p]),
    "bpr_train_stream_batched": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                         c_int32, c_float, c_uint64, c_uint64, c_int64, c_void_p]),
    "bpr_shuffle_epoch": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_uint64, c_void_p,
                                  c_void_p]),
    "bpr_train_strict": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32,
                                 c_float, c_uint64, c_uint64, c_int64, c_void_p]),
    "bpr_item_delta": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "bpr_item_fold": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int32, c_int64,
                              c_void_p]),
    "bpr_set_stream_opts": (c_int, [c_void_p, c_int32, c_int32]),
    "bpr_stream_run_len": (c_int, [c_void_p]),
    "bpr_auc_rows": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bpr_set_hot_lds": (c_int, [c_void_p, c_int32, c_int32]),
    "bpr_stream_lds_rows": (c_int, [c_void_p]),
    "bpr_item_fold_delta": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int64,
                                    c_void_p]),
    "bpr_set_hot_rows": (c_int, [c_void_p, c_int32, c_int32]),
    "bpr_set_heavy_users": (c_int, [c_void_p, c_int32, c_int64]),
    "bpr_set_hot_items": (c_int, [c_void_p, c_void_p, c_int32, c_void_p]),
    "bpr_hot_rows": (c_int, [c_void_p, c_void_p]),
    "bpr_hot_tier_begin": (c_int, [c_void_p, c_void_p]),
    "bpr_hot_exchange": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "bpr_hot_tier_end": (c_int, [c_void_p]),
    "bpr_sync_cut": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_float,
                             c_int32]),
    "bpr_plan_epoch": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_uint64, c_void_p,
                               c_void_p]),
    "bpr_plan_chunk": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_uint64, c_int64, c_void_p,
                               c_void_p, c_int32]),
    "bpr_flush_lazy": (c_int, [c_void_p]),
    "bpr_flush_items": (c_int, [c_void_p]),
    "bpr_get_step_host": (c_int, [c_void_p, POINTER(c_int64)]),
    "bpr_set_step": (c_int, [c_void_p, c_int64]),
    "bpr_set_sampler_iter": (c_int, [c_void_p, c_int64]),
    "bpr_timing_enable": (c_int, [c_void_p, c_int32]),
    "bpr_timing_read_host": (c_int, [c_void_p, POINTER(c_double), POINTER(c_int64)]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load libbprcore.so and declare every prototype.  Raises if the library is missing."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is not built. "
                "Run __graft_entry__.build() (or `make -C revisit-bpr_amd/csrc`). "
                "There is no CPU fallback."
            )
        # torch ships its own HIP runtime; load it FIRST so libbprcore's libamdhip64 dependency
        # resolves to the same runtime instance (two runtimes in one process cannot share a GPU)
        import torch  # noqa: F401

        lib = ctypes.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != OK:
        raise BprError(rc, (load().bpr_last_error() or b"").decode("utf-8", "replace"))
