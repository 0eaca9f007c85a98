"""JSON-lines interaction files ⇄ device-ready arrays.

On-disk formats are the reference's (bin/datasets/format-repro.sh:56-81, jsonl.sh:77-83):
  full-train-with-fold-in.jsonl                  {"user": u, "item": i}            one per line
  full-train-with-fold-in-user-seen-items.jsonl  {"user": u, "seen_items": [...]}  one per user
  test-grouped.jsonl                             {"user": u, "item": [...]}        one per eval user
The reference parses them with json.loads per line into a scipy dok matrix (minutes on MSD,
experiments/bpr/dataset.py:183-190); here the native loader (include/bprio.h → libbprio.so,
`native_io`: mmap, one piece of the file per thread, C++ CSR builder) produces the arrays the
engine consumes.  `reader="pyarrow"` is a second, independent implementation (pyarrow's JSON reader +
numpy) kept for cross-checks.
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np

from revisit_bpr.datasets.synthetic import Interactions

TRAIN, SEEN, TEST = ("full-train-with-fold-in.jsonl",
                     "full-train-with-fold-in-user-seen-items.jsonl", "test-grouped.jsonl")


def _read_table(path: Path):
    import pyarrow.json as paj

    return paj.read_json(str(path))


def _ragged(table, key: str):
    """(users, indptr, flat values) of a {"user": u, key: [...]} file."""
    users = table.column("user").to_numpy()
    col = table.column(key).combine_chunks()
    offsets = col.offsets.to_numpy().astype(np.int64)
    flat = col.values.to_numpy().astype(np.int64)
    return users.astype(np.int64), offsets, flat


def load_dataset(path, num_users: int, num_items: int, reader: str = "native",
                 threads: int = 0) -> Interactions:
    """Read the three files of a dataset directory.  Duplicate (user, item) pairs are dropped, as
    the reference's dok matrix does."""
    path = Path(path)
    if reader == "native":
        return _load_native(path, num_users, num_items, threads)
    if reader != "pyarrow":
        raise ValueError("reader must be 'native' or 'pyarrow'")
    t = _read_table(path / TRAIN)
    u = t.column("user").to_numpy().astype(np.int64)
    i = t.column("item").to_numpy().astype(np.int64)
    if u.size and (u.max() >= num_users or i.max() >= num_items or min(u.min(), i.min()) < 0):
        raise ValueError("ids out of range for the given num_users / num_items")
    key = np.unique(u * num_items + i)
    u, i = key // num_items, key % num_items
    # seen items → CSR (sorted, de-duplicated, item 0 dropped)
    su, soff, sflat = _ragged(_read_table(path / SEEN), "seen_items")
    rows = np.repeat(su, np.diff(soff))
    skey = np.unique(rows * num_items + sflat)
    skey = skey[skey % num_items != 0]
    indptr = np.zeros(num_users + 1, np.int64)
    np.add.at(indptr, skey // num_items + 1, 1)
    indptr = np.cumsum(indptr)
    indices = (skey % num_items).astype(np.int32)
    ev_users = np.zeros(0, np.int32)
    ev_ptr, ev_items = np.zeros(1, np.int64), np.zeros(0, np.int32)
    if (path / TEST).exists():
        eu, eoff, eflat = _ragged(_read_table(path / TEST), "item")
        ev_users, ev_ptr, ev_items = eu.astype(np.int32), eoff, eflat.astype(np.int32)
    return Interactions(num_users=num_users, num_items=num_items, users=u.astype(np.int32),
                        items=i.astype(np.int32), indptr=indptr, indices=indices,
                        eval_users=ev_users, eval_indptr=ev_ptr, eval_items=ev_items)


def _load_native(path: Path, num_users: int, num_items: int, threads: int) -> Interactions:
    from revisit_bpr.datasets import native_io as nio

    try:
        tu, ti = nio.read_pairs(path / TRAIN, "item", threads)
        # de-duplicated training pairs in (user, item) order = the rows of their CSR, expanded
        tptr, tidx = nio.build_csr(tu, ti, num_users, num_items, drop_item0=False, threads=threads)
        su, soff, sflat = nio.read_ragged(path / SEEN, "seen_items", threads)
        rows = np.repeat(su, np.diff(soff))
        indptr, indices = nio.build_csr(rows, sflat, num_users, num_items, drop_item0=True,
                                        threads=threads)
    except nio.BprIoError as e:
        if "out of range" in str(e):
            raise ValueError(str(e)) from e
        raise
    users = np.repeat(np.arange(num_users, dtype=np.int32), np.diff(tptr))
    ev_users = np.zeros(0, np.int32)
    ev_ptr, ev_items = np.zeros(1, np.int64), np.zeros(0, np.int32)
    if (path / TEST).exists():
        ev_users, ev_ptr, ev_items = nio.read_ragged(path / TEST, "item", threads)
    return Interactions(num_users=num_users, num_items=num_items, users=users, items=tidx,
                        indptr=indptr, indices=indices, eval_users=ev_users, eval_indptr=ev_ptr,
                        eval_items=ev_items)


def write_dataset(data: Interactions, path) -> None:
    """Write `data` in the reference's three-file JSONL layout (used for synthetic end-to-end runs)."""
    path = Path(path)
    path.mkdir(parents=True, exist_ok=True)
    with (path / TRAIN).open("w") as fh:
        fh.write("".join(f'{{"user": {int(u)}, "item": {int(i)}}}\n'
                         for u, i in zip(data.users, data.items)))
    with (path / SEEN).open("w") as fh:
        for u in range(1, data.num_users):
            row = data.indices[data.indptr[u]:data.indptr[u + 1]]
            fh.write(json.dumps({"user": u, "seen_items": [int(x) for x in row]}) + "\n")
    with (path / TEST).open("w") as fh:
        for k, u in enumerate(data.eval_users):
            row = data.eval_items[data.eval_indptr[k]:data.eval_indptr[k + 1]]
            if len(row):  # the reference's test files list only users with held-out items
                fh.write(json.dumps({"user": int(u), "item": [int(x) for x in row]}) + "\n")
