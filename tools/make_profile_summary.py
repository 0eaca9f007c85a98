#!/usr/bin/env python
"""Turn the rocprofv3 outputs under gpurun_out/ (kernel stats CSV, FETCH_SIZE / WRITE_SIZE counter
collections of bench.py, calibration runs of tools/ubench/pmc_calib) into the committed summaries
under profiles/ and the traffic JSON that bench.py reports.   python tools/make_profile_summary.py r01"""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
G, P = ROOT / "gpurun_out", ROOT / "profiles"
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def agg(path):
    a = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = (r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])
        a[k][0] += 1
        a[k][1] += float(r["Counter_Value"])
    return a


shutil.copy(G / f"prof_{tag}" / "bench_kernel_stats.csv", P / f"{tag}_bench_kernel_stats.csv")
if (G / f"prof_{tag}_sync" / "bench_kernel_stats.csv").exists():
    shutil.copy(G / f"prof_{tag}_sync" / "bench_kernel_stats.csv", P / f"{tag}_bench_sync_kernel_stats.csv")
if (G / f"prof_{tag}_adam" / "bench_kernel_stats.csv").exists():
    shutil.copy(G / f"prof_{tag}_adam" / "bench_kernel_stats.csv", P / f"{tag}_bench_adam_yelp_kernel_stats.csv")
out = [f"# {tag} — rocprofv3 PMC passes (separate runs): FETCH_SIZE and WRITE_SIZE, unit KB (x1024 B)",
       "# command: rocprofv3 --pmc <COUNTER> --output-format csv -- python bench.py --steps 24 --warmup 4 --no-cpu-baseline   (tools/profile_round.sh)",
       "", "## calibration on known byte counts (tools/ubench/pmc_calib.hip, 1 GiB buffers > Infinity Cache)",
       "kernel,counter,calls,per_call_KB,true_KB,ratio"]
true = {"k_read_dword": 1048576, "k_read_dwordx4": 1048576, "k_gather_rows": 1048576,
        "k_write_dword": 1048576, "k_atomic_dword": 131072}
for name in ("calib_fetch", "calib_write"):
    f = G / name / "calib_counter_collection.csv"
    if not f.exists():
        continue
    for (k, c), (n, v) in sorted(agg(f).items()):
        kk = k.split("<")[0]
        if kk in true:
            t = true[kk] if ((c == "FETCH_SIZE") == kk.startswith(("k_read", "k_gather"))) else 0
            out.append(f"{kk},{c},{n},{v / n:.1f},{t},{(v / n / t if t else float('nan')):.3f}")
out += ["",
        "=> on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes read for dword-per-lane, dwordx4-per-lane and 512-B row gathers alike",
        "   (MI355X_MICROARCH.md: FETCH_SIZE = TCC_EA0_RDREQ x 64 B with 128-B requests tallied at 64 B) -> corrected read bytes = 2 x FETCH_SIZE x 1024;",
        "   WRITE_SIZE is exact for plain stores AND for fp32 atomics (an atomic line request is counted as its bytes written).",
        "", "## bench.py kernels, per launch", "kernel,counter,calls,per_call_KB"]
res = {}
for name in ("pmc_fetch", "pmc_write"):
    for (k, c), (n, v) in sorted(agg(G / name / "bench_counter_collection.csv").items(), key=lambda x: -x[1][1]):
        out.append(f"{k},{c},{n},{v / n:.1f}")
        if "k_stream<" in k:
            res[c] = v / n
rd, wr = 2 * res["FETCH_SIZE"] * 1024, res["WRITE_SIZE"] * 1024
alg = 199168 * 3080
out += ["", f"k_stream per launch (199,168 triples): read {rd / 1e6:.1f} MB (corrected), written {wr / 1e6:.1f} MB, "
            f"total {(rd + wr) / 1e6:.1f} MB;",
        f"algorithmic 24d+8 = 3080 B/triple -> {alg / 1e6:.1f} MB; traffic/algorithmic = {(rd + wr) / alg:.3f}",
        "(reads: 3 rows + order/CSR lookups; writes are BELOW 3 rows/triple because a user row is written once per user-run, not per triple)"]
(P / f"{tag}_pmc_traffic.md").write_text("\n".join(out) + "\n")
json.dump({"round": tag, "kernel": "k_stream<32,4,ADAPTIVE,bitmap>", "workload": "ml-20m d=128 adaptive",
           "triples_per_launch": 199168, "fetch_size_kb_raw": res["FETCH_SIZE"], "write_size_kb": res["WRITE_SIZE"],
           "read_bytes_corrected": rd, "write_bytes": wr, "traffic_bytes_per_launch": rd + wr,
           "correction": "read = 2 x FETCH_SIZE x 1024 (gfx950, calibrated in profiles/%s_pmc_traffic.md); write = WRITE_SIZE x 1024" % tag},
          open(P / f"traffic_{tag}.json", "w"), indent=1)
# BASELINE configs[3]: MSD shape, d = 256 (k_stream<64,4,...>), when its passes were collected
if (G / "pmc_fetch_msd" / "bench_counter_collection.csv").exists():
    rm = {}
    for name in ("pmc_fetch_msd", "pmc_write_msd"):
        for (k, c), (n, v) in agg(G / name / "bench_counter_collection.csv").items():
            if "k_stream<" in k:
                rm[c] = v / n
    chunk_msd = int(41141 * __import__("math").log(41141) / 256) * 256
    rd, wr = 2 * rm["FETCH_SIZE"] * 1024, rm["WRITE_SIZE"] * 1024
    alg = chunk_msd * (24 * 256 + 8)
    with open(P / f"{tag}_pmc_traffic.md", "a") as f:
        f.write(f"\n## MSD shape, d = 256 (BASELINE configs[3]): k_stream per launch ({chunk_msd} triples): read "
                f"{rd / 1e6:.1f} MB (corrected), written {wr / 1e6:.1f} MB, total {(rd + wr) / 1e6:.1f} MB; "
                f"algorithmic 24d+8 = 6152 B/triple -> {alg / 1e6:.1f} MB; traffic/algorithmic = {(rd + wr) / alg:.3f}\n")
    json.dump({"round": tag, "kernel": "k_stream<64,4,ADAPTIVE,bitmap>", "workload": "msd d=256 adaptive",
               "triples_per_launch": chunk_msd, "fetch_size_kb_raw": rm["FETCH_SIZE"], "write_size_kb": rm["WRITE_SIZE"],
               "read_bytes_corrected": rd, "write_bytes": wr, "traffic_bytes_per_launch": rd + wr,
               "correction": "as traffic_%s.json" % tag}, open(P / f"traffic_{tag}_msd_d256.json", "w"), indent=1)
print(out[-3], out[-2], sep="\n")
