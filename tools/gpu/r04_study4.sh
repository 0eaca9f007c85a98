#!/bin/bash
# r4 cadence study, part 4: chunks below period / N at lr 0.05 (budget 4,000 -> 2.5 N chunks per period)
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_study; mkdir -p $O
S="timeout 1500 python tools/cadence_study.py --cadence auto --lr 0.05 --seeds 10"
$S --epochs 4 --ranks 1,2,4,8 --hot-rows 1024 > $O/lr05_auto4N_H1024.txt 2>&1
$S --epochs 8 --eval-every 2 --ranks 1,8 --hot-rows 1024 > $O/lr05_auto4N_H1024_8ep.txt 2>&1
$S --epochs 4 --ranks 1,8 --hot-rows 0 > $O/lr05_auto4N_H0.txt 2>&1
timeout 1500 python tools/cadence_study.py --cadence auto --lr 0.0094 --epochs 20 --eval-every 5 --seeds 10 --ranks 1,4,8 --hot-rows 0 > $O/lr0094_auto_H0.txt 2>&1
grep -h "^#" $O/lr05_auto4N_*.txt $O/lr0094_auto_H0.txt | cut -c1-400
