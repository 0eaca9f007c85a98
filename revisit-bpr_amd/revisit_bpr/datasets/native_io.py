"""ctypes binding of the native JSONL loader (include/bprio.h → revisit-bpr_amd/libbprio.so).

Host-side IO for the reference's on-disk formats (bin/datasets/format-repro.sh:56-81,
jsonl.sh:77-83): files are mapped, cut at line boundaries into one piece per thread and scanned
once; `build_csr` turns (user, item) pairs into the sorted, de-duplicated seen-items CSR that
`bpr_bind_seen_csr` consumes.  The reference does this with json.loads per line into a scipy dok
matrix (experiments/bpr/dataset.py:183-190).
"""
from __future__ import annotations

import ctypes
from ctypes import POINTER, byref, c_char_p, c_int, c_int32, c_int64, c_void_p
from pathlib import Path

import numpy as np

_LIB_PATH = Path(__file__).resolve().parents[2] / "libbprio.so"
_lib = None

SIGNATURES = {
    "bprio_version": (c_int, []),
    "bprio_last_error": (c_char_p, []),
    "bprio_free": (None, [c_void_p]),
    "bprio_read_pairs": (c_int, [c_char_p, c_char_p, c_int, POINTER(c_void_p), POINTER(c_void_p),
                                 POINTER(c_int64)]),
    "bprio_read_ragged": (c_int, [c_char_p, c_char_p, c_int, POINTER(c_void_p), POINTER(c_void_p),
                                  POINTER(c_void_p), POINTER(c_int64), POINTER(c_int64)]),
    "bprio_build_csr": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p,
                                POINTER(c_void_p), POINTER(c_int64)]),
}


class BprIoError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise BprIoError(f"{_LIB_PATH} is missing: build it with `make -C revisit-bpr_amd/csrc` "
                             "(or python -c 'import __graft_entry__ as g; g.build()')")
        lib = ctypes.CDLL(str(_LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _check(lib, rc: int) -> None:
    if rc != 0:
        raise BprIoError(lib.bprio_last_error().decode())


def _take(lib, ptr: c_void_p, n: int, dtype) -> np.ndarray:
    """copy a malloc'ed buffer into a numpy array and release it"""
    out = np.empty(n, dtype)
    if n:
        ctypes.memmove(out.ctypes.data, ptr.value, out.nbytes)
    lib.bprio_free(ptr)
    return out


def read_pairs(path, key: str = "item", threads: int = 0):
    """{"user": u, key: i} lines → (users int32 [n], values int32 [n]) in file order."""
    lib = load()
    pu, pv, n = c_void_p(), c_void_p(), c_int64()
    _check(lib, lib.bprio_read_pairs(str(path).encode(), key.encode(), threads, byref(pu), byref(pv),
                                     byref(n)))
    return _take(lib, pu, n.value, np.int32), _take(lib, pv, n.value, np.int32)


def read_ragged(path, key: str, threads: int = 0):
    """{"user": u, key: [...]} lines → (users int32 [rows], offsets int64 [rows+1], values int32)."""
    lib = load()
    pu, po, pv, rows, nv = c_void_p(), c_void_p(), c_void_p(), c_int64(), c_int64()
    _check(lib, lib.bprio_read_ragged(str(path).encode(), key.encode(), threads, byref(pu), byref(po),
                                      byref(pv), byref(rows), byref(nv)))
    return (_take(lib, pu, rows.value, np.int32), _take(lib, po, rows.value + 1, np.int64),
            _take(lib, pv, nv.value, np.int32))


def build_csr(users: np.ndarray, items: np.ndarray, num_users: int, num_items: int,
              drop_item0: bool = True, threads: int = 0):
    """(user, item) pairs → (indptr int64 [num_users+1], indices int32) sorted and de-duplicated
    per row."""
    lib = load()
    users = np.ascontiguousarray(users, np.int32)
    items = np.ascontiguousarray(items, np.int32)
    if users.shape != items.shape:
        raise ValueError("users / items differ in length")
    indptr = np.empty(num_users + 1, np.int64)
    pidx, nnz = c_void_p(), c_int64()
    _check(lib, lib.bprio_build_csr(users.ctypes.data, items.ctypes.data, users.size, num_users,
                                    num_items, int(drop_item0), threads, indptr.ctypes.data,
                                    byref(pidx), byref(nnz)))
    return indptr, _take(lib, pidx, nnz.value, np.int32)
