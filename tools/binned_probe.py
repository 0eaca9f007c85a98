#!/usr/bin/env python
"""How do the columns of a TRAINED item table take k_sort_binned's equi-depth binning?  Trains the ML-20M shape
(bench.py's data, lr and schedule-free STREAM epochs), then — per checkpoint — emulates the kernel's binning in
numpy (same float32 arithmetic) to count the columns whose largest bin exceeds BIN_MAX (they fall back to the
radix sort) and times one refresh with the binned and with the radix sort on the idle chip.
    python tools/binned_probe.py [lr] [epochs,epochs,...]"""
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "revisit-bpr_amd"))
from revisit_bpr.datasets import synthetic  # noqa: E402
from revisit_bpr.fast import StreamTrainer  # noqa: E402
from revisit_bpr.models import BPR  # noqa: E402
from revisit_bpr.models.bpr import MF  # noqa: E402

lr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.001
marks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 10, 30, 60]
BINS, BIN_MAX = 8192, 64


def emulate(Q):
    """largest bin per column, as k_sort_binned classifies (float32 where the kernel uses float32)"""
    I, d = Q.shape
    worst = np.zeros(d, np.int64)
    for f in range(d):
        col = Q[:, f]
        if col.max() <= col.min():
            worst[f] = I
            continue
        cmax = np.float32(col.max())
        cscale = np.float32(np.float32(1024.0) / (np.float32(col.max()) - np.float32(col.min())))
        x = (cmax - col) * cscale
        cb = np.clip(x.astype(np.int32), 0, 1023)
        coarse = np.bincount(cb, minlength=1024)
        cum = np.concatenate([[0], np.cumsum(coarse)[:-1]])
        frac = np.clip(x - cb.astype(np.float32), np.float32(0), np.float32(0.999))
        r = cum[cb].astype(np.float32) + frac * coarse[cb].astype(np.float32)
        crowded = np.nonzero(coarse > 32)[0]
        if len(crowded):  # the second level over the crowded stretch
            h_lo, h_hi = int(crowded[0]), int(crowded[-1])
            inside = (cb >= h_lo) & (cb <= h_hi)
            ftop = np.float32(cmax - np.float32(h_lo) / cscale)
            fscale = np.float32(cscale * np.float32(1024.0 / (h_hi - h_lo + 1)))
            x2 = (ftop - col) * fscale
            fb = np.clip(x2.astype(np.int32), 0, 1023)
            fine = np.bincount(fb[inside], minlength=1024)
            fcum = np.concatenate([[0], np.cumsum(fine)[:-1]])
            frac2 = np.clip(x2 - fb.astype(np.float32), np.float32(0), np.float32(0.999))
            r2 = np.float32(cum[h_lo]) + (fcum[fb].astype(np.float32) + frac2 * fine[fb].astype(np.float32))
            r = np.where(inside, r2, r)
        bins = np.clip((r * np.float32(BINS / I)).astype(np.int32), 0, BINS - 1)
        worst[f] = np.bincount(bins, minlength=BINS).max()
    return worst


data = synthetic.generate_latent(136677, 20108, 9_700_000, factors=16, strength=1.2, median_per_user=37,
                                 min_per_user=5, seed=13, eval_users=10_000, item_skew=1.2, item_shift=60.0,
                                 cache_dir=tempfile.gettempdir())
t = {k: torch.from_numpy(getattr(data, k)).cuda() for k in ("users", "items", "indptr", "indices")}
torch.manual_seed(13)
model = BPR(fuse_forward=True, reg_alphas={"user": 0.0016, "item": 0.0001, "neg": 0.00375},
            logits_model=MF(torch.nn.Embedding(data.num_users, 128, padding_idx=0),
                            torch.nn.Embedding(data.num_items, 128, padding_idx=0))).cuda()
tr = StreamTrainer(model, t["users"], t["items"], t["indptr"], t["indices"], lr=lr, sampler="adaptive",
                   adaptive_p=0.01, batch_size=256, seed=1)
e = model.engine()
done = 0
print(f"lr {lr}")
for m in marks:
    for _ in range(m - done):
        tr.train_epoch()
    done = m
    Q = model.logits_model.get_features()["item"].data.cpu().numpy()
    worst = emulate(Q)
    times = {}
    for binned in (0, 1):
        e.set_tuning("binned_sort", binned)
        for _ in range(3):
            e.adaptive_refresh()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            e.adaptive_refresh()
        b.record()
        torch.cuda.synchronize()
        times[binned] = a.elapsed_time(b) / 20 * 1000
    e.set_tuning("binned_sort", 1)
    print(f"after epoch {m}: largest bin per column: median {int(np.median(worst))} p90 {int(np.percentile(worst, 90))} "
          f"max {worst.max()}; columns over {BIN_MAX}: {(worst > BIN_MAX).sum()} of {len(worst)} (over 32: "
          f"{(worst > 32).sum()}, over 16: {(worst > 16).sum()}); refresh {times[1]:.1f} us binned, {times[0]:.1f} us radix",
          flush=True)
