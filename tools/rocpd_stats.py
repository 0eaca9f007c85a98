#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace) as a per-kernel stats table.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/<name>.txt
"""
import re
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in c.execute(f"pragma table_info({disp})")]
    scols = [r[1] for r in c.execute(f"pragma table_info({sym})")]
    namecol = "kernel_name" if "kernel_name" in scols else "display_name"
    rows = c.execute(
        f"select s.{namecol}, count(*), sum(d.end - d.start), min(d.end - d.start), "
        f"max(d.end - d.start) from {disp} d join {sym} s on d.kernel_id = s.id "
        f"group by s.{namecol} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# {path}\n# columns of dispatch table: {cols}")
    print(f"{'kernel':<90} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}")
    for name, n, tot, mn, mx in rows:
        short = re.sub(r"\s+", " ", name)[:88]
        print(f"{short:<90} {n:>7} {tot / 1e6:>10.3f} {tot / n / 1e3:>10.2f} {mn / 1e3:>9.2f} "
              f"{mx / 1e3:>9.2f} {100.0 * tot / total:>6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
