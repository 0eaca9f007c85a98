#!/bin/bash
# Everything profiles/ is built from, in one GPU-box call:  bash tools/profile_round.sh r03
# (then, back in the build container: python tools/make_profile_summary.py r03)
T=${1:-r04}
R=/root/repo; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_final.log 2>&1; tail -1 $O/bench_final.log > $O/bench_${T}_final.json
python $R/bench.py --steps 20 --warmup 5 > $O/bench_driverlike.log 2>&1; tail -1 $O/bench_driverlike.log > $O/bench_${T}_driverlike.json
rm -rf $O/prof_$T $O/pmc_fetch $O/pmc_write $O/prof_${T}_adam $O/calib_fetch $O/calib_write $O/prof_${T}_sync
# the bench configuration (overlapped snapshot schedule) and the reference's schedule beside it
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o bench -- python $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline > $O/prof_${T}_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}_sync -o bench -- python $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline --refresh-lag 0 > $O/prof_${T}_sync.log 2>&1
# HBM-side traffic of the dominant kernel: separate --pmc passes
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- python $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline --sustained-epochs 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- python $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline --sustained-epochs 0 > /dev/null 2>&1
# the same two passes for BASELINE configs[3] (MSD shape, d = 256: the HBM-bound case)
rm -rf $O/pmc_fetch_msd $O/pmc_write_msd
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_msd -o bench -- python $R/bench.py --workload msd --dim 256 --steps 12 --warmup 4 --no-cpu-baseline --sustained-epochs 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_msd -o bench -- python $R/bench.py --workload msd --dim 256 --steps 12 --warmup 4 --no-cpu-baseline --sustained-epochs 0 > /dev/null 2>&1
# calibration of the two counters on known byte counts
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -o calib -- $R/tools/ubench/pmc_calib > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/calib_write -o calib -- $R/tools/ubench/pmc_calib > /dev/null 2>&1
# BASELINE configs[4]: the Adam path (batched STREAM) on the Yelp shape
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}_adam -o bench -- python $R/bench.py --workload yelp --optimizer adam --steps 24 --warmup 30 --no-cpu-baseline > $O/prof_${T}_adam.log 2>&1
find $O/prof_$T $O/prof_${T}_sync $O/prof_${T}_adam $O/pmc_fetch $O/pmc_write $O/calib_fetch $O/calib_write -name "*.csv" | head -30
tail -1 $O/bench_final.log | cut -c1-300
grep -h "^{" $O/prof_${T}_bench.log $O/prof_${T}_sync.log $O/prof_${T}_adam.log | cut -c1-220
python $R/tools/timeline.py $(find $O/prof_$T -name "*kernel_trace.csv" | head -1) 0 | tail -4
