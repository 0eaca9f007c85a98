"""Datasets and eval collators named by the reference's BPR configs
(experiments/bpr/dataset.py of the reference: SparseSamplingInMemoryWithCollator :142-190,
InMemory :16-33, Iter :36-54, OnePosCollator :193-225, ManyPosCollator :228-271,
AllItemsCollator :274-304) — same constructor arguments and batch keys.

The training dataset additionally exposes the seen-items CSR (``seen_csr()``) so the experiment can
bind it to the engine once instead of shipping a padded [B, max_seen] matrix every batch.
"""
from __future__ import annotations

import json
from itertools import islice
from pathlib import Path
from typing import Any, Iterator

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence
from torch.utils.data import Dataset, IterableDataset, get_worker_info


def _lines(path) -> Iterator[dict]:
    with Path(path).open("r", encoding="utf-8") as fh:
        for line in fh:
            yield json.loads(line)


def _seen_map(path) -> dict[int, list[int]]:
    return {row["user"]: row["seen_items"] for row in _lines(path)}


def _columns(instances: list[dict[str, Any]]) -> dict[str, list]:
    cols: dict[str, list] = {}
    for inst in instances:
        for key, value in inst.items():
            cols.setdefault(key, []).append(value)
    return cols


class InMemory(Dataset):
    """Eval samples {"user", "item": [...]} joined with the user's seen items."""

    def __init__(self, path: Path | str, seen_items_path: Path | str) -> None:
        self._samples = list(_lines(path))
        self._seen = _seen_map(seen_items_path)

    def __len__(self) -> int:
        return len(self._samples)

    def __getitem__(self, idx: int) -> dict[str, Any]:
        sample = self._samples[idx]
        return {**sample, "seen_items": self._seen[sample["user"]]}


class Iter(IterableDataset):
    def __init__(self, path: Path | str, seen_items_path: Path | str) -> None:
        self._path = Path(path)
        self._seen = _seen_map(seen_items_path)

    def __iter__(self) -> Iterator[dict[str, Any]]:
        info = get_worker_info()
        first, stride = (info.id, info.num_workers) if info is not None and info.num_workers > 0 \
            else (0, 1)
        with self._path.open("r", encoding="utf-8") as fh:
            for line in islice(fh, first, None, stride):
                sample = json.loads(line)
                sample["seen_items"] = self._seen[sample["user"]]
                yield sample


class SparseSamplingInMemoryWithCollator(Dataset):
    """Training interactions as index tensors; ``__getitem__`` returns the index and
    ``collate_fn(indices)`` gathers (user, item, seen_items) — optionally resident on the GPU."""

    def __init__(self, path: Path | str, seen_items_path: Path | str, num_users: int,
                 num_items: int, padding_value: float = 0, put_on_cuda: bool = False) -> None:
        from revisit_bpr.datasets.interactions import _ragged, _read_table

        t = _read_table(Path(path))
        u = t.column("user").to_numpy().astype(np.int64)
        i = t.column("item").to_numpy().astype(np.int64)
        key = np.unique(u * num_items + i)  # (user, item) pairs de-duplicated, user-major order
        self._user_ids = torch.from_numpy(key // num_items)
        self._item_ids = torch.from_numpy(key % num_items)
        su, soff, sflat = _ragged(_read_table(Path(seen_items_path)), "seen_items")
        rows = np.repeat(su, np.diff(soff))
        skey = np.unique(rows * num_items + sflat)
        skey = skey[skey % num_items != 0]
        counts = np.bincount(skey // num_items, minlength=num_users)
        self._indptr = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int64))
        self._indices = torch.from_numpy((skey % num_items).astype(np.int32))
        # the padded matrix the reference's samplers index per batch ([U, max_seen], 0 = pad)
        width = max(int(counts.max()) if counts.size else 0, 1)
        pad = np.full((num_users, width), padding_value, np.int64)
        col = np.arange(skey.size) - np.repeat(np.cumsum(counts) - counts, counts)
        pad[skey // num_items, col] = skey % num_items
        self._seen_items = torch.from_numpy(pad)
        if put_on_cuda and torch.cuda.is_available():
            for name in ("_user_ids", "_item_ids", "_seen_items", "_indptr", "_indices"):
                setattr(self, name, getattr(self, name).cuda())

    def __len__(self) -> int:
        return len(self._user_ids)

    def __getitem__(self, idx: int) -> int:
        return idx

    def collate_fn(self, indices: list[int]) -> dict[str, torch.Tensor]:
        idx = torch.as_tensor(indices, device=self._user_ids.device)
        users = self._user_ids[idx]
        return {"user": users, "item": self._item_ids[idx], "seen_items": self._seen_items[users]}

    def seen_csr(self) -> tuple[torch.Tensor, torch.Tensor]:
        return self._indptr, self._indices


def _unseen(num_items: int, seen) -> torch.Tensor:
    keep = torch.ones(num_items, dtype=torch.bool)
    keep[0] = False
    keep[torch.as_tensor(seen, dtype=torch.long).view(-1)] = False
    return torch.arange(num_items)[keep]


class OnePosCollator:
    """Leave-one-out eval row: column 0 = the held-out positive (``item`` indexes into the user's
    seen list), then every unseen item; target marks column 0."""

    def __init__(self, num_items: int) -> None:
        self._num_items = num_items

    def __call__(self, instances: list[dict[str, Any]]) -> dict[str, torch.Tensor]:
        batch = {k: torch.tensor(v) for k, v in _columns(instances).items()}
        seen = batch["seen_items"].view(-1)
        positive = seen[batch["item"]]
        batch["item"] = torch.hstack((positive.unsqueeze(0),
                                      _unseen(self._num_items, seen).unsqueeze(0)))
        target = torch.zeros_like(batch["item"], dtype=torch.float)
        target[:, 0] = 1.0
        batch["target"] = target
        return batch


class ManyPosCollator:
    """Eval rows = the user's held-out positives followed by every unseen item, padded."""

    def __init__(self, num_items: int, padding_value: float = 0) -> None:
        self._num_items = num_items
        self._padding_value = padding_value

    def __call__(self, instances: list[dict[str, Any]]) -> dict[str, torch.Tensor]:
        cols = _columns(instances)
        items, targets = [], []
        for pos, seen in zip(cols["item"], cols["seen_items"]):
            row = torch.hstack((torch.tensor(pos), _unseen(self._num_items, seen)))
            tgt = torch.zeros_like(row, dtype=torch.float)
            tgt[:len(pos)] = 1.0
            items.append(row)
            targets.append(tgt)
        pad = functools_pad(self._padding_value)
        out = {"user": torch.as_tensor(cols["user"]), "item": pad(items),
               "seen_items": pad([torch.as_tensor(s) for s in cols["seen_items"]]),
               "target": pad(targets)}
        out["mask"] = out["item"].gt(self._padding_value).float()
        return out


def functools_pad(value):
    return lambda seqs: pad_sequence(seqs, batch_first=True, padding_value=value)


class AllItemsCollator:
    """Eval rows score EVERY item: item = arange(num_items), target marks the held-out positives."""

    def __init__(self, num_items: int, padding_value: float = 0) -> None:
        self._num_items = num_items
        self._padding_value = padding_value

    def __call__(self, instances: list[dict[str, Any]]) -> dict[str, torch.Tensor]:
        cols = _columns(instances)
        n = len(cols["user"])
        target = torch.zeros(n, self._num_items)
        for r, pos in enumerate(cols["item"]):
            target[r, torch.as_tensor(pos, dtype=torch.long)] = 1.0
        return {
            "user": torch.as_tensor(cols["user"]),
            "item": torch.arange(self._num_items, dtype=torch.long).expand(n, -1).contiguous(),
            "target": target,
            "seen_items": functools_pad(self._padding_value)(
                [torch.as_tensor(s) for s in cols["seen_items"]]),
        }
