#!/bin/bash
# r6: everything the round's profiles are built from, one GPU-box call:  bash profiles/r06_scripts/r06_profile.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_final.log 2>&1; tail -1 $O/bench_final.log > $O/bench_r06_1gpu.json
python $R/bench.py --steps 20 --warmup 5 > $O/bench_driverlike.log 2>&1; tail -1 $O/bench_driverlike.log > $O/bench_r06_1gpu_driverlike.json
# kernel trace + stats of the bench command on the TRAINED state (31st and 32nd epoch timed), tier on and off
for v in lds nolds; do
  extra=""; [ $v = nolds ] && extra="--hot-lds 0"
  rm -rf $O/prof_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o bench -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --sustained-epochs 1 --steady-epochs 30 --steady-timed-epochs 2 $extra > $O/prof_$v.log 2>&1
done
# memory-side counters on the trained state, separate --pmc passes, tier on and off
for v in lds nolds; do
  extra=""; [ $v = nolds ] && extra="--hot-lds 0"
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_ATOMIC TCC_ATOMIC" "TCC_EA0_RDREQ TCC_REQ TCC_HIT TCC_MISS" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"; do
    i=$((i+1)); rm -rf $O/pmc_${v}_$i
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_${v}_$i -o x -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --sustained-epochs 0 --steady-epochs 30 --steady-timed-epochs 1 $extra > $O/pmc_${v}_$i.log 2>&1
  done
done
python - <<PY | tee $O/r06_pmc_trained_state.txt
import csv, glob, collections
print("# r06: k_stream on the TRAINED state (the 31st epoch's 47 launches of bench.py ... --steady-epochs 30 --steady-timed-epochs 1), per launch")
for v in ("nolds", "lds"):
    agg = collections.OrderedDict()
    for d in sorted(glob.glob("$O/pmc_%s_*" % v)):
        for f in glob.glob(d + "/**/x_counter_collection.csv", recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if "k_stream<" in r["Kernel_Name"]]
            by = collections.defaultdict(list)
            for r in rows:
                by[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            for c, vals in by.items():
                vals.sort()
                last = [x for _, x in vals[-47:]]
                first = [x for _, x in vals[:47]]
                agg[c] = (sum(last) / len(last), sum(first) / len(first), len(vals))
    for c, (a, b, n) in agg.items():
        print("%-6s %-28s trained %16.1f   first epoch %16.1f   (%d launches seen)" % (v, c, a, b, n))
PY
python - <<PY | tee $O/r06_kernel_trained_state.txt
import csv, glob, collections
for v in ("nolds", "lds"):
    f = glob.glob("$O/prof_%s/**/*kernel_trace.csv" % v, recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    ks = [r for r in rows if "k_stream<" in r["Kernel_Name"]]
    last = ks[-94:]
    t0, t1 = int(last[0]["Start_Timestamp"]), int(last[-1]["End_Timestamp"])
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s >= t0 and e <= t1:
            k = r["Kernel_Name"].split("(")[0][:80]
            agg[k][0] += 1; agg[k][1] += e - s
    print("# %s: epochs 31-32 (94 launches) under rocprofv3 --kernel-trace; first epoch's k_stream avg %.1f us" % (v, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks[:47]) / 47e3))
    for k, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:7]:
        print("%s, %d, %.1f us" % (k, n, ns / n / 1e3))
    print("# span %.3f ms = %.4f ms per step" % ((t1 - t0) / 1e6, (t1 - t0) / 1e6 / 94))
PY
find $O -name "*kernel_stats.csv" | head; tail -1 $O/bench_final.log | cut -c1-400
