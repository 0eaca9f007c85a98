#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_g; mkdir -p $O
python tools/e2e_many_seeds.py adam uniform 200 1 strict-own-order,batched > $O/many_adam_uniform.txt 2>&1; grep -v amdgpu $O/many_adam_uniform.txt | cut -c1-230
timeout 2400 python tools/cadence_study.py --cadence auto --hot-rows 1024 --lr 0.001 --epochs 160 --eval-every 40 --seeds 6 --ranks 1,8 > $O/lr001_auto_H1024_fused.txt 2>&1; grep "^#" $O/lr001_auto_H1024_fused.txt | cut -c1-400
