#!/usr/bin/env python
"""Per-rank cost of the reconciliation passes of a multi-rank step, measured on ONE GPU (the parts of
DESIGN.md §7's table that do not need a second device): the hot-tier exchange pass, the cold tier's
fused fold + delta pass, the un-fused snapshot cut (k_transpose) and the sort — HIP events, ML-20M
shape, d = 128.    python tools/time_sync_parts.py [hot_rows ...]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "revisit-bpr_amd")]
import torch  # noqa: E402

from revisit_bpr import engine as eng  # noqa: E402
from revisit_bpr.distributed import ItemSync, LocalWorld  # noqa: E402


def timed(fn, reps=50):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        fn()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    dev = torch.device("cuda")
    U, I, d = 136678, 20109, 128
    g = torch.Generator(device=dev).manual_seed(1)
    counts = (torch.rand(I, generator=g, device=dev) ** 8 * 1e5).long().cpu()
    for H in [int(x) for x in (sys.argv[1:] or ["256", "1024", "4096"])]:
        lw = LocalWorld(2)
        es = [eng.Engine(torch.zeros(U, d, device=dev), torch.randn(I, d, device=dev) * 0.01) for _ in range(2)]
        ss = [ItemSync([es[r].Q], comm=lw.member(r), engine=es[r], hot_rows=H, item_counts=counts) for r in range(2)]
        e, s = es[0], ss[0]
        t_hot = timed(lambda: e.hot_exchange(s._hb, s._htot, True, True, s.base[0]))
        lib = s._lib
        t_cold = timed(lambda: lib.bpr_item_fold_delta(e.Q.data_ptr(), s.base[0].data_ptr(), s._own[0].data_ptr(),
                                                       s._tot[0].data_ptr(), 1.0, e.Q.numel(),
                                                       torch.cuda.current_stream().cuda_stream))
        e.adaptive_refresh()
        t_refresh = timed(lambda: e.adaptive_refresh(), reps=20)
        print(f"H {H:5d}: hot exchange pass {t_hot:6.1f} us ({H * d * 4 / 1024:.0f} KB message), cold fold+delta pass "
              f"{t_cold:6.1f} us ({I * d * 4 / 1e6:.1f} MB message), full refresh (cut + sort, whole chip) {t_refresh:6.1f} us")
        for x in ss:
            x.close()


if __name__ == "__main__":
    main()
