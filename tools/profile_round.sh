#!/bin/bash
# Everything profiles/ is built from, in one GPU-box call:  bash tools/profile_round.sh r03
# (then, back in the build container: python tools/make_profile_summary.py r03)
T=${1:-r05}
R=/root/repo; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_final.log 2>&1; tail -1 $O/bench_final.log > $O/bench_${T}_final.json
python $R/bench.py --steps 20 --warmup 5 > $O/bench_driverlike.log 2>&1; tail -1 $O/bench_driverlike.log > $O/bench_${T}_driverlike.json
rm -rf $O/prof_$T $O/pmc_fetch $O/pmc_write $O/prof_${T}_adam $O/calib_fetch $O/calib_write $O/prof_${T}_sync
# the bench configuration (overlapped snapshot schedule) and the reference's schedule beside it
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o bench -- python $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline --steady-epochs 0 > $O/prof_${T}_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}_sync -o bench -- python $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline --refresh-lag 0 --steady-epochs 0 > $O/prof_${T}_sync.log 2>&1
# the trained state: the same command run on to 30 epochs; the summary below is over the LAST epoch's launches
rm -rf $O/prof_${T}_steady
rocprofv3 --kernel-trace --output-format csv -d $O/prof_${T}_steady -o bench -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --sustained-epochs 0 --steady-epochs 30 > $O/prof_${T}_steady.log 2>&1
# HBM-side traffic of the dominant kernel: separate --pmc passes
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- python $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline --sustained-epochs 0 --steady-epochs 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- python $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline --sustained-epochs 0 --steady-epochs 0 > /dev/null 2>&1
# the same two passes for BASELINE configs[3] (MSD shape, d = 256: the HBM-bound case)
rm -rf $O/pmc_fetch_msd $O/pmc_write_msd
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_msd -o bench -- python $R/bench.py --workload msd --dim 256 --steps 12 --warmup 4 --no-cpu-baseline --sustained-epochs 0 --steady-epochs 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_msd -o bench -- python $R/bench.py --workload msd --dim 256 --steps 12 --warmup 4 --no-cpu-baseline --sustained-epochs 0 --steady-epochs 0 > /dev/null 2>&1
# calibration of the two counters on known byte counts
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -o calib -- $R/tools/ubench/pmc_calib > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/calib_write -o calib -- $R/tools/ubench/pmc_calib > /dev/null 2>&1
# BASELINE configs[4]: the Adam path (batched STREAM) on the Yelp shape
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}_adam -o bench -- python $R/bench.py --workload yelp --steps 24 --warmup 30 --no-cpu-baseline --steady-epochs 0 > $O/prof_${T}_adam.log 2>&1
find $O/prof_$T $O/prof_${T}_sync $O/prof_${T}_adam $O/pmc_fetch $O/pmc_write $O/calib_fetch $O/calib_write -name "*.csv" | head -30
tail -1 $O/bench_final.log | cut -c1-300
grep -h "^{" $O/prof_${T}_bench.log $O/prof_${T}_sync.log $O/prof_${T}_adam.log | cut -c1-220
python $R/tools/timeline.py $(find $O/prof_$T -name "*kernel_trace.csv" | head -1) 0 | tail -4
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
