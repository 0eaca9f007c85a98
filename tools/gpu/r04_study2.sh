#!/bin/bash
# r4 cadence study, part 2: the reference configs' learning rates (0.0094: 20 epochs; 0.001: 160 epochs)
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_study; mkdir -p $O
S="timeout 1500 python tools/cadence_study.py --lr 0.0094 --epochs 20 --eval-every 5 --seeds 10"
$S --ranks 1,2,4,8 --cadence job --hot-rows 0 > $O/lr0094_job.txt 2>&1
$S --ranks 1,2,4,8 --cadence rank --hot-rows 0 > $O/lr0094_rank_H0.txt 2>&1
$S --ranks 1,2,4,8 --cadence rank --hot-rows 1024 > $O/lr0094_rank_H1024_s1.txt 2>&1
$S --ranks 1,4,8 --cadence rank --hot-rows 1024 --hot-split 4 > $O/lr0094_rank_H1024_s4.txt 2>&1
$S --ranks 1,4,8 --cadence rank --hot-rows 4096 --hot-split 4 > $O/lr0094_rank_H4096_s4.txt 2>&1
grep -h "^#" $O/lr0094_*.txt | cut -c1-400
S="timeout 2400 python tools/cadence_study.py --lr 0.001 --epochs 160 --eval-every 40 --seeds 6"
$S --ranks 1,4,8 --cadence rank --hot-rows 1024 > $O/lr001_rank_H1024_s1.txt 2>&1
$S --ranks 1,4,8 --cadence rank --hot-rows 0 > $O/lr001_rank_H0.txt 2>&1
$S --ranks 1,4,8 --cadence job --hot-rows 0 > $O/lr001_job.txt 2>&1
grep -h "^#" $O/lr001_*.txt | cut -c1-400
S="timeout 1200 python tools/cadence_study.py --lr 0.05 --epochs 4 --seeds 10"
$S --ranks 1,4,8 --cadence rank --hot-rows 0 --cold-scale mean > $O/lr05_rank_H0_mean.txt 2>&1
grep -h "^#" $O/lr05_rank_H0_mean.txt | cut -c1-400
