#!/bin/bash
# Everything profiles/ is built from, in one GPU-box call:  bash tools/profile_round.sh
# (then, back in the build container: python tools/make_profile_summary.py r01)
R=/root/repo; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_final.log 2>&1; tail -1 $O/bench_final.log > $O/bench_r01_final.json
rm -rf $O/prof_r01 $O/pmc_fetch $O/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r01 -o bench -- python $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline > $O/prof_r01_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- python $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- python $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline > /dev/null 2>&1
find $O/prof_r01 $O/pmc_fetch $O/pmc_write -name "*.csv" | head -20
tail -1 $O/prof_r01_bench.log | cut -c1-300
