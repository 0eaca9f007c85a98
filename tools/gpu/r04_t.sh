#!/bin/bash
# the model with the reference's optional item_bias, beside the model without (bench.py --item-bias)
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_t; mkdir -p $O
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" > $O/$tag.log 2>&1; tail -1 $O/$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['value']/1e6,1), 'M triples/s', round(d['ms_per_step'],4), 'ms/step, roofline', round(d['roofline']['frac'],4), 'item_bias', d['config']['item_bias'])" 2>&1 | tee -a $O/summary.txt; }
for b in 0 1 0 1; do
  run ml20m_sgd_bias$b --steps 100 --warmup 10 --sustained-epochs 0 --item-bias $b
done
for b in 0 1; do
  run yelp_adam_bias$b --workload yelp --optimizer adam --warmup 30 --steps 24 --item-bias $b
  run yelp_momentum_bias$b --workload yelp --optimizer momentum --warmup 30 --steps 24 --item-bias $b
  BPR_VS_DIRECT=1 run yelp_adam_direct1_bias$b --workload yelp --optimizer adam --warmup 30 --steps 24 --item-bias $b
done
