#!/usr/bin/env python
"""Assemble profiles/r04_cadence_study.txt from the summary lines of tools/cadence_study.py runs
(gpurun_out/r04_study/*.txt)."""
import glob
import re
import sys

HEAD = """# r4 multi-rank cadence study (VERDICT r3 item 1) — tools/cadence_study.py on ONE MI355X.
# N ranks of the PRODUCT trainer (fast.StreamTrainer + distributed.ItemSync: user shards, replicated item table,
# every sum folded one protocol step after it was cut) stepped round-robin in one process over
# distributed.LocalWorld; full ML-20M-shaped latent set (136,677 x 20,108, 9.55 M train triples), d = 128,
# adaptive sampler p = 0.01, L2 (0.0016, 0.0001, 0.00375), snapshot schedule lag 1 (what bench.py times).
# Epochs per learning rate: lr 0.05: 4; lr 0.0094: 20 (evaluated every 5); lr 0.001: 160 (every 40).
# Columns: cadence (job = period / N triples per rank and chunk; rank = a full period per rank; auto = the
# staleness budget lr x N x chunk <= 4,000: 1 .. 4 N chunks per period; 'auto(<=N)' = its first version, capped at
# period / N), H = rows of the hot tier (0 = one tier), s = launches per chunk with a
# hot exchange after each, world = ranks; nDCG@100 per evaluated epoch (seed mean), dnDCG = difference to the
# 1-rank runs of the same invocation.
"""


def main():
    rows = []
    for f in sorted(glob.glob("gpurun_out/r04_study/lr*.txt")):
        for line in open(f):
            if line.startswith("# lr"):
                line = line[2:].rstrip()
                if "auto4N" in f:  # the shipped rule: up to 4 N chunks per period
                    line = line.replace("cadence auto", "cadence auto(<=4N)")
                elif "autofused" in f:  # the shipped trainer: reconciliation passes + cut as one pass (bpr_sync_cut)
                    line = line.replace("cadence auto", "cadence auto(fused pass)")
                elif "lr05_auto_H" in f:  # the first version of the rule: at most N chunks (= job cadence)
                    line = line.replace("cadence auto", "cadence auto(<=N)")
                rows.append(line)
    key = lambda r: (float(re.search(r"lr ([0-9.]+)", r).group(1)), r.split(" | ")[0], int(re.search(r"world (\d+)", r).group(1)))
    rows.sort(key=key)
    out = [HEAD]
    last = None
    for r in rows:
        cfg = r.split(" | ")[0]
        if cfg != last:
            out.append("")
            last = cfg
        out.append(r)
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
