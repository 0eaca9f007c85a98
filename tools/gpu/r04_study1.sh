#!/bin/bash
# r4 cadence study, part 1: lr 0.05 (4 epochs) — where does the hot tier have to be exchanged how often?
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_study; mkdir -p $O
S="timeout 1200 python tools/cadence_study.py --lr 0.05 --epochs 4 --seeds 10"
$S --ranks 1,2,4,8 --cadence job --hot-rows 0 > $O/lr05_job.txt 2>&1
for H in 1024 4096; do
  $S --ranks 1,2 --cadence rank --hot-rows $H --hot-split 1 > $O/lr05_rank_H${H}_s1.txt 2>&1
  $S --ranks 1,2,4 --cadence rank --hot-rows $H --hot-split 2 > $O/lr05_rank_H${H}_s2.txt 2>&1
  $S --ranks 1,4,8 --cadence rank --hot-rows $H --hot-split 4 > $O/lr05_rank_H${H}_s4.txt 2>&1
  $S --ranks 1,4,8 --cadence rank --hot-rows $H --hot-split 8 > $O/lr05_rank_H${H}_s8.txt 2>&1
  $S --ranks 1,8 --cadence rank --hot-rows $H --hot-split 16 > $O/lr05_rank_H${H}_s16.txt 2>&1
done
$S --ranks 1,4,8 --cadence rank --hot-rows 256 --hot-split 8 > $O/lr05_rank_H256_s8.txt 2>&1
$S --ranks 1,8 --cadence rank --hot-rows 16384 --hot-split 8 > $O/lr05_rank_H16384_s8.txt 2>&1
grep -h "^#" $O/lr05_*.txt | cut -c1-330
