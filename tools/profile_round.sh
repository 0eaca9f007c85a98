#!/bin/bash
# Everything profiles/ is built from, in one GPU-box call:  bash tools/profile_round.sh r02
# (then, back in the build container: python tools/make_profile_summary.py r02)
T=${1:-r02}
R=/root/repo; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_final.log 2>&1; tail -1 $O/bench_final.log > $O/bench_${T}_final.json
rm -rf $O/prof_$T $O/pmc_fetch $O/pmc_write $O/prof_${T}_adam
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o bench -- python $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline > $O/prof_${T}_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- python $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- python $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline > /dev/null 2>&1
# BASELINE configs[4]: the Adam path (batched STREAM) on the Yelp shape
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}_adam -o bench -- python $R/bench.py --workload yelp --optimizer adam --steps 24 --warmup 4 --no-cpu-baseline > $O/prof_${T}_adam.log 2>&1
find $O/prof_$T $O/prof_${T}_adam $O/pmc_fetch $O/pmc_write -name "*.csv" | head -20
tail -1 $O/prof_${T}_bench.log | cut -c1-300
tail -1 $O/prof_${T}_adam.log | cut -c1-300
