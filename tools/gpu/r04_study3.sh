#!/bin/bash
# r4 cadence study, part 3: the staleness-budget cadence ("auto", budget 4,000) validated; sync parts timed
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_study; mkdir -p $O
python tools/time_sync_parts.py > $O/sync_parts.txt 2>&1; cat $O/sync_parts.txt
S="timeout 1500 python tools/cadence_study.py --cadence auto --hot-rows 1024"
$S --lr 0.0094 --epochs 20 --eval-every 5 --seeds 10 --ranks 1,2,4,8 > $O/lr0094_auto_H1024.txt 2>&1
$S --lr 0.05 --epochs 4 --seeds 10 --ranks 1,2,4,8 > $O/lr05_auto_H1024.txt 2>&1
$S --lr 0.05 --epochs 8 --eval-every 2 --seeds 10 --ranks 1,8 --hot-rows 0 > $O/lr05_auto_H0_8ep.txt 2>&1
grep -h "^#" $O/lr0094_auto_H1024.txt $O/lr05_auto_H1024.txt $O/lr05_auto_H0_8ep.txt | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_multirank_parity.py -x -q -k "check_rccl or c_abi" > $O/pytest_rccl.txt 2>&1; tail -5 $O/pytest_rccl.txt
