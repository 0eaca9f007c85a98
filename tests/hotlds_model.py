"""A numpy restatement of the LDS tier's ALGORITHM (k_stream<..., LDSHOT>, revisit-bpr_amd/csrc/bpr_stream.h, and
its launcher in csrc/bprcore.hip) — test infrastructure:

  * the partition of a launch's triples into runs: [0, tail1) in runs of L, [tail1, tail2) in runs of L / 2,
    [tail2, n) in runs of L / 4, every zone whole wave-loads of runs (`zones`, `runs_of`: the host's and the
    kernel's integer arithmetic side by side);
  * the deal of run pairs to persistent workgroups: ticket k of workgroup b = wave-load k x grid + b (`deal`);
  * the algebra of the private delta blocks: a workgroup reads a hot row as Q + its OWN delta, adds its updates to
    that delta, and the deltas are summed into the table at the launch's end; cold rows are updated in place
    (`run_launch`: plain SGD on one embedding dimension per row is enough to pin the algebra).

The HIP kernel itself is held to the oracle in tests/test_gpu_hotlds.py (`-m gpu`)."""
import numpy as np


def zones(n: int, L: int, gpw: int, tail_percent: int):
    """(tail1, tail2) as launch_stream computes them: zones hold whole wave-loads of runs; what is left of the chunk
    past the last whole wave-load of full runs always goes in the shortest runs."""
    len2 = max(1, L // 2)
    wl = L * gpw
    t1 = int(n * (1.0 - tail_percent / 100.0)) // wl * wl
    t2 = t1 + int((n - t1) * 0.6) // (len2 * gpw) * (len2 * gpw)
    if tail_percent <= 0:
        t1 = t2 = n // wl * wl
    return t1, t2


def runs_of(n: int, L: int, tail1: int, tail2: int):
    """[(t0, t1)] of every run, in run order, as the kernel derives them from (run index, tail1, tail2)."""
    L2, L3 = (L >> 1 if L >= 2 else 1), (L >> 2 if L >= 4 else 1)
    R1 = tail1 // L
    R2 = R1 + (tail2 - tail1) // L2
    n_runs = R2 + (n - tail2 + L3 - 1) // L3
    out = []
    for run in range(n_runs):
        if run >= R2:
            Lr, t0 = L3, tail2 + (run - R2) * L3
        elif run >= R1:
            Lr, t0 = L2, tail1 + (run - R1) * L2
        else:
            Lr, t0 = L, run * L
        out.append((t0, min(t0 + Lr, n)))
    return out, R1, R2


def deal(n_runs: int, grid: int, gpw: int):
    """runs of every workgroup in the order its tickets hand them out: ticket k -> runs (k * grid + b) * gpw + {0..gpw-1}"""
    per = [[] for _ in range(grid)]
    for b in range(grid):
        k = 0
        while (k * grid + b) * gpw < n_runs:
            base = (k * grid + b) * gpw
            per[b].extend(r for r in range(base, min(base + gpw, n_runs)))
            k += 1
    return per


def run_launch(q0: np.ndarray, rows: np.ndarray, grads, grid: int, gpw: int, L: int, tail_percent: int, hot: np.ndarray,
               order: str = "round-robin", seed: int = 0):
    """One launch over triples whose item row is rows[t]: every triple adds grads(value it reads) to its row.  Hot
    rows (hot[row] True) take a workgroup's updates in its private delta and are read as q + own delta; at the end
    the deltas are summed into q.  The workgroups' runs are interleaved in time by `order`.  Returns (q, reads)."""
    n = len(rows)
    t1, t2 = zones(n, L, gpw, tail_percent)
    runs, _, _ = runs_of(n, L, t1, t2)
    per = deal(len(runs), grid, gpw)
    q = q0.astype(np.float64).copy()
    delta = np.zeros((grid, len(q)))
    reads = np.full(n, np.nan)
    cursor = [0] * grid
    rng = np.random.default_rng(seed)
    alive = [b for b in range(grid) if per[b]]
    while alive:
        b = alive[0] if order == "one-by-one" else (alive[int(rng.integers(len(alive)))] if order == "random" else alive[0])
        if order == "round-robin":
            alive.append(alive.pop(0))
            b = alive[-1]
        t0, t1_ = runs[per[b][cursor[b]]]
        for t in range(t0, t1_):
            r = rows[t]
            v = q[r] + (delta[b, r] if hot[r] else 0.0)
            reads[t] = v
            g = grads(v, t)
            if hot[r]:
                delta[b, r] += g
            else:
                q[r] += g
        cursor[b] += 1
        if cursor[b] == len(per[b]):
            alive.remove(b)
    q += delta.sum(0)  # the flush + the epilogue's fold
    return q, reads
