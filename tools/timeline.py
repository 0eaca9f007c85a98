#!/usr/bin/env python
"""Kernel timeline of a rocprofv3 --kernel-trace CSV: per kernel the average duration, and for the
steady-state loop the gaps between consecutive kernels of each queue (= stream) and the overlap of
the refresh kernels with k_stream.

    python tools/timeline.py gpurun_out/prof/.../bench_kernel_trace.csv [first_fraction_to_skip]
"""
import csv
import sys
from collections import defaultdict


def short(n):
    for key in ("k_stream_epilogue_cut", "k_stream_epilogue", "k_stream", "k_sort_sub", "k_merge_runs",
                "k_transpose", "k_plan", "DeviceRadixSort", "radix", "k_sum_partials"):
        if key in n:
            return key
    return n.split("(")[0][-40:]


def main(path, skip=0.3):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                         r.get("Queue_Id", "0"), r.get("Stream_Id", r.get("Queue_Id", "0"))))
    rows.sort()
    t0, t1 = rows[0][0], rows[-1][1]
    rows = [r for r in rows if r[0] >= t0 + skip * (t1 - t0)]
    dur = defaultdict(list)
    for s, e, n, q, st in rows:
        dur[n].append(e - s)
    print("kernel                      calls   avg_us")
    for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f"{n:26s} {len(v):6d} {sum(v) / len(v) / 1e3:8.2f}")
    ks = [r for r in rows if r[2] == "k_stream"]
    if len(ks) > 3:
        per = [(ks[i + 1][0] - ks[i][0]) / 1e3 for i in range(len(ks) - 1)]
        per.sort()
        print(f"k_stream start-to-start: median {per[len(per) // 2]:.2f} us  (p10 {per[len(per) // 10]:.2f}, p90 {per[9 * len(per) // 10]:.2f})")
        # what sits between the end of one k_stream and the start of the next, on any queue
        gaps = []
        for i in range(len(ks) - 1):
            a, b = ks[i][1], ks[i + 1][0]
            inside = [r for r in rows if r[0] >= a and r[1] <= b and r[2] != "k_stream"]
            busy = sum(r[1] - r[0] for r in inside)
            gaps.append(((b - a) / 1e3, busy / 1e3, [r[2] for r in inside]))
        gaps.sort(key=lambda g: g[0])
        g = gaps[len(gaps) // 2]
        print(f"between two k_stream launches: median {g[0]:.2f} us, of which kernels {g[1]:.2f} us: {g[2]}")
        # overlap of sort kernels with k_stream
        ov = tot = 0
        for s, e, n, q, st in rows:
            if n in ("k_sort_sub", "k_merge_runs"):
                tot += e - s
                for ks_s, ks_e, *_ in ks:
                    ov += max(0, min(e, ks_e) - max(s, ks_s))
        if tot:
            print(f"sort kernels: {100.0 * ov / tot:.1f} % of their time runs beside a k_stream launch")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.3)
