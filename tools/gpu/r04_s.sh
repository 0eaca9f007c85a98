#!/bin/bash
# rocprofv3 kernel stats of BASELINE configs[4] (Yelp shape, Adam) with the r4 k_vstream
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_s; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_adam
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_adam -o bench -- python $R/bench.py --workload yelp --optimizer adam --steps 24 --warmup 30 --no-cpu-baseline > $O/prof_adam.log 2>&1
find $O/prof_adam -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/adam_yelp_kernel_stats.csv
head -8 $O/adam_yelp_kernel_stats.csv | cut -c1-180
grep -h "^{" $O/prof_adam.log | cut -c1-300
python $R/bench.py --workload yelp --optimizer adam --steps 24 --warmup 30 2>/dev/null | tail -1 > $O/bench_yelp_adam.json; cut -c1-200 $O/bench_yelp_adam.json
