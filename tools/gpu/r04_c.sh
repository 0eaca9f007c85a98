#!/bin/bash
# r4 call 3: two-tier tests + a first look at the cadence study (lr 0.05, 3 seeds)
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r04_c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_two_tier.py tests/test_metrics_product.py tests/test_gpu_parity.py -x -q -m gpu -k "two_tier or local_world or metric or evaluate or cut or split" > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt
tail -5 $O/pytest.txt
S="timeout 1500 python tools/cadence_study.py --lr 0.05 --epochs 4 --seeds 3"
$S --ranks 1,2,4,8 --cadence rank --hot-rows 1024 > $O/look_rank_H1024.txt 2>&1
$S --ranks 4,8 --cadence rank --hot-rows 0 > $O/look_rank_H0.txt 2>&1
$S --ranks 4,8 --cadence rank --hot-rows 4096 --hot-split 4 > $O/look_rank_H4096_s4.txt 2>&1
$S --ranks 4,8 --cadence rank --hot-rows 256 --hot-split 4 > $O/look_rank_H256_s4.txt 2>&1
$S --ranks 8 --cadence job --hot-rows 0 > $O/look_job.txt 2>&1
grep -h "^#\|Error\|error" $O/look_*.txt | cut -c1-400
grep -h '"s":' $O/look_rank_H1024.txt | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print(j['world'], j['seed'], j['s'], j['replica_spread'])"
