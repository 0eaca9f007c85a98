#!/bin/bash
# r5, second half (binned sort, sorter on 32 CUs): the bench lines and rocprofv3 summaries profiles/ is built from.
# A trimmed tools/profile_round.sh: the MSD / Adam / calibration passes are unchanged kernels (profiles/r05_*).
T=${1:-r05b}
R=/root/repo; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_final.log 2>&1; tail -1 $O/bench_final.log > $O/bench_${T}_final.json
python $R/bench.py --steps 20 --warmup 5 > $O/bench_driverlike.log 2>&1; tail -1 $O/bench_driverlike.log > $O/bench_${T}_driverlike.json
rm -rf $O/prof_$T $O/pmc_fetch $O/pmc_write $O/prof_${T}_sync $O/prof_${T}_steady
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o bench -- python $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline --steady-epochs 0 > $O/prof_${T}_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}_sync -o bench -- python $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline --refresh-lag 0 --steady-epochs 0 > $O/prof_${T}_sync.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/prof_${T}_steady -o bench -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --sustained-epochs 0 --steady-epochs 30 > $O/prof_${T}_steady.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- python $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline --sustained-epochs 0 --steady-epochs 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- python $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline --sustained-epochs 0 --steady-epochs 0 > /dev/null 2>&1
for f in $(find $O/prof_$T $O/prof_${T}_sync -name "*kernel_stats.csv"); do cp $f $O/$(basename $(dirname $(dirname $f)))_kernel_stats.csv 2>/dev/null; done
find $O/prof_$T $O/prof_${T}_sync $O/pmc_fetch $O/pmc_write -name "*.csv" | head -20
tail -1 $O/bench_final.log | cut -c1-400
python $R/tools/timeline.py $(find $O/prof_$T -name "*kernel_trace.csv" | head -1) 0 | tail -6 | tee $O/${T}_timeline.txt
python - <<PY | tee $O/${T}_steady_last_epoch.txt
import csv, glob, collections
f = glob.glob("$O/prof_${T}_steady/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [r for r in rows if "k_stream<" in r["Kernel_Name"]]
last = ks[-47:]
t0, t1 = int(last[0]["Start_Timestamp"]), int(last[-1]["End_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= t0 and e <= t1:
        k = r["Kernel_Name"].split("(")[0][:70]
        agg[k][0] += 1; agg[k][1] += e - s
print("# ${T}: the LAST epoch (47 launches) of bench.py --steps 8 --warmup 4 --sustained-epochs 0 --steady-epochs 30 under rocprofv3 --kernel-trace")
print("# kernel, calls, avg_us   (first epoch of the same trace: k_stream avg %.1f us over its first 47 launches)" % (sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks[:47]) / 47e3))
for k, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:6]:
    print("%s, %d, %.1f" % (k, n, ns / n / 1e3))
print("# epoch span %.3f ms = %.4f ms per step" % ((t1 - t0) / 1e6, (t1 - t0) / 1e6 / 47))
PY
