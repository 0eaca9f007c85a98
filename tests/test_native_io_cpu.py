"""The native JSONL loader (include/bprio.h → libbprio.so) against an independent reading of the
same files (json.loads per line, the reference's way: experiments/bpr/dataset.py:183-190) and
against the pyarrow path.  CPU only."""
import json
import re
from pathlib import Path

import numpy as np
import pytest

from revisit_bpr.datasets import interactions, native_io, synthetic

ROOT = Path(__file__).resolve().parent.parent


def test_header_exports_and_ctypes_table_agree():
    declared = set(re.findall(r"\b(bprio_[a-z_]+)\s*\(", (ROOT / "include" / "bprio.h").read_text()))
    assert declared == set(native_io.SIGNATURES)
    lib = native_io.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.bprio_version() >= 100


def _by_json(path, key):
    rows = [json.loads(line) for line in Path(path).read_text().splitlines() if line.strip()]
    return [r["user"] for r in rows], [r[key] for r in rows]


@pytest.mark.parametrize("threads", [1, 3, 0])
def test_three_files_round_trip(tmp_path, threads):
    data = synthetic.generate(400, 150, 9000, median_per_user=12, seed=4, eval_users=60)
    interactions.write_dataset(data, tmp_path)
    u, i = native_io.read_pairs(tmp_path / interactions.TRAIN, "item", threads)
    ju, ji = _by_json(tmp_path / interactions.TRAIN, "item")
    assert np.array_equal(u, np.array(ju, np.int32)) and np.array_equal(i, np.array(ji, np.int32))
    for fname, key in ((interactions.SEEN, "seen_items"), (interactions.TEST, "item")):
        ru, off, val = native_io.read_ragged(tmp_path / fname, key, threads)
        ju, jv = _by_json(tmp_path / fname, key)
        assert np.array_equal(ru, np.array(ju, np.int32))
        assert np.array_equal(np.diff(off), [len(v) for v in jv])
        assert np.array_equal(val, np.array([x for v in jv for x in v], np.int32))
    a = interactions.load_dataset(tmp_path, data.num_users, data.num_items, reader="native",
                                  threads=threads)
    b = interactions.load_dataset(tmp_path, data.num_users, data.num_items, reader="pyarrow")
    for f in ("users", "items", "indptr", "indices", "eval_users", "eval_indptr", "eval_items"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
        assert getattr(a, f).dtype == getattr(b, f).dtype, f
    assert np.array_equal(a.indptr, data.indptr) and np.array_equal(a.indices, data.indices)


def test_build_csr_sorts_dedups_and_drops_the_pad_item():
    rng = np.random.default_rng(3)
    U, I, n = 50, 40, 3000
    u = rng.integers(0, U, n).astype(np.int32)
    i = rng.integers(0, I, n).astype(np.int32)
    for drop in (True, False):
        indptr, idx = native_io.build_csr(u, i, U, I, drop_item0=drop, threads=2)
        want = [sorted({int(b) for a, b in zip(u, i) if a == r and not (drop and b == 0)}) for r in range(U)]
        assert np.array_equal(np.diff(indptr), [len(w) for w in want])
        assert np.array_equal(idx, np.array([x for w in want for x in w], np.int32))
    with pytest.raises(native_io.BprIoError):
        native_io.build_csr(np.array([U], np.int32), np.array([1], np.int32), U, I)
    indptr, idx = native_io.build_csr(np.zeros(0, np.int32), np.zeros(0, np.int32), U, I)
    assert indptr.tolist() == [0] * (U + 1) and idx.size == 0


def test_parser_edge_cases(tmp_path):
    f = tmp_path / "a.jsonl"
    f.write_text('{"item": 7, "user": 3}\n'               # keys in any order
                 '  { "user" :4 ,"item":  9 }  \r\n'        # whitespace, CRLF
                 '\n'                                      # blank line
                 '{"user": 5, "ts": "2020-01-01, x}", "tags": ["a", "b]"], "w": 1.5e3, "item": 11}\n'
                 '{"user": 6, "item": 2147483647}')        # no trailing newline, largest id
    u, i = native_io.read_pairs(f, "item", 2)
    assert u.tolist() == [3, 4, 5, 6] and i.tolist() == [7, 9, 11, 2147483647]
    g = tmp_path / "b.jsonl"
    g.write_text('{"user": 1, "seen_items": []}\n{"user": 2, "seen_items": [5]}\n'
                 '{"seen_items": [ 1 , 2,3 ], "user": 3}\n{"user": 4, "seen_items": 8}\n')
    ru, off, val = native_io.read_ragged(g, "seen_items", 1)
    assert ru.tolist() == [1, 2, 3, 4] and off.tolist() == [0, 0, 1, 4, 5] and val.tolist() == [5, 1, 2, 3, 8]
    empty = tmp_path / "e.jsonl"
    empty.write_text("")
    u, i = native_io.read_pairs(empty, "item")
    assert u.size == 0 and i.size == 0
    for bad in ('{"user": 1}\n', '{"user": -1, "item": 2}\n', '{"user": 1, "item": [2]}\n',
                '{"user": 1, "item": 2147483648}\n', 'not json\n', '{"user": 1, "item": 2\n'):
        h = tmp_path / "bad.jsonl"
        h.write_text('{"user": 1, "item": 1}\n' + bad)
        with pytest.raises(native_io.BprIoError, match="cannot parse"):
            native_io.read_pairs(h, "item")
    with pytest.raises(native_io.BprIoError, match="cannot open"):
        native_io.read_pairs(tmp_path / "missing.jsonl", "item")


def test_large_file_many_threads(tmp_path):
    """a few MB, so that the file really is cut into several pieces at line boundaries"""
    rng = np.random.default_rng(0)
    n = 300_000
    u = rng.integers(1, 50_000, n)
    i = rng.integers(1, 20_000, n)
    f = tmp_path / "big.jsonl"
    f.write_text("".join(f'{{"user": {a}, "item": {b}}}\n' for a, b in zip(u, i)))
    assert f.stat().st_size > 4 * (1 << 20)
    for threads in (1, 4, 8):
        gu, gi = native_io.read_pairs(f, "item", threads)
        assert np.array_equal(gu, u) and np.array_equal(gi, i)
