#!/usr/bin/env python
"""Time of one adaptive-snapshot refresh (cut + sort) on the idle chip: k_sort_binned against the radix sort.
    python tools/sort_time.py [I d]          (default: the ML-20M shape)"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "revisit-bpr_amd"))
from revisit_bpr.engine import Engine  # noqa: E402

shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(20109, 128), (17771, 64), (4801, 64), (20109, 64)]
for I, d in shapes:
    rng = np.random.default_rng(1)
    for kind in ("random-init", "trained-like"):
        Q = (rng.standard_normal((I, d)) * 0.05).astype(np.float32)
        if kind == "trained-like":
            cold = rng.random(I) < 0.6
            Q[cold] *= 0.02
            Q[:, : d // 4] *= 5.0
        Q[0] = 0
        for binned in (0, 1):
            e = Engine(torch.zeros(4, d, device="cuda"), torch.from_numpy(Q).cuda(), None)
            e.set_tuning("binned_sort", binned)
            for _ in range(5):
                e.adaptive_refresh()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50):
                e.adaptive_refresh()
            b.record()
            torch.cuda.synchronize()
            print(f"I={I} d={d} {kind:12s} binned={binned}: {a.elapsed_time(b) / 50 * 1000:.1f} us per refresh", flush=True)
