"""JSONL datasets + padding collator behind the reference's API (revisit_bpr/datasets/jsonl.py).
On-disk format: one JSON object per line, e.g. {"user": u, "item": i} /
{"user": u, "seen_items": [...]} / {"user": u, "item": [...]} (bin/datasets/*.sh of the reference)."""
from __future__ import annotations

import json
from itertools import islice
from pathlib import Path
from typing import Any, Iterator

import torch
from torch.nn.utils.rnn import pad_sequence
from torch.utils.data import Dataset, IterableDataset, get_worker_info


def _read_lines(path) -> list[dict]:
    with Path(path).open("r", encoding="utf-8") as fh:
        return [json.loads(line) for line in fh]


class InMemory(Dataset):
    """All samples of a JSON-lines file, parsed up front."""

    def __init__(self, path: Path | str) -> None:
        self._samples = _read_lines(path)

    def __len__(self) -> int:
        return len(self._samples)

    def __getitem__(self, idx: int) -> dict[str, Any]:
        return self._samples[idx]


class Iter(IterableDataset):
    """Streams a JSON-lines file; DataLoader workers take interleaved lines."""

    def __init__(self, path: Path | str) -> None:
        self._path = Path(path)

    def __iter__(self) -> Iterator[dict[str, Any]]:
        info = get_worker_info()
        first, stride = (info.id, info.num_workers) if info is not None and info.num_workers > 0 \
            else (0, 1)
        with self._path.open("r", encoding="utf-8") as fh:
            for line in islice(fh, first, None, stride):
                yield json.loads(line)


class Collator:
    """Stack a list of sample dicts; keys in `pad` are right-padded with `padding_value` and get a
    `<key>_mask` float companion."""

    def __init__(self, pad: list[str] | None = None, padding_value: float = 0) -> None:
        self._pad = set(pad or [])
        self._padding_value = padding_value

    def __call__(self, instances: list[dict[str, Any]]) -> dict[str, torch.Tensor]:
        columns: dict[str, list] = {}
        for inst in instances:
            for key, value in inst.items():
                columns.setdefault(key, []).append(value)
        batch = {}
        for key, values in columns.items():
            if key in self._pad:
                batch[key] = pad_sequence([torch.as_tensor(v) for v in values], batch_first=True,
                                          padding_value=self._padding_value)
            else:
                batch[key] = torch.tensor(values)
        for key in self._pad:
            batch[f"{key}_mask"] = batch[key].ne(self._padding_value).float()
        return batch
