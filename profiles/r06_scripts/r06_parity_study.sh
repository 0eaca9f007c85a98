#!/bin/bash
# r6: does the LDS tier of the hot block keep the curves?  (tools/fullepoch_study.py; ML-20M shape, d = 128, adaptive)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_parity; mkdir -p $O
# lr 0.01 (inside the budget by 0.4 %): exact mini-batches vs the timed schedule without / with the tier, 8 seeds
LR=0.01 EPOCHS=3,4,6,8 NSEEDS=8 timeout 900 python tools/fullepoch_study.py strict timed_nolds timed_lds reference_lds > $O/lr0.01.txt 2>&1
tail -6 $O/lr0.01.txt
# lr 0.05 (outside the budget): the reference's own loop (fixture) vs the reference's schedule with the tier forced on
NSEEDS=6 timeout 600 python tools/fullepoch_study.py reference_nolds reference_lds > $O/lr0.05.txt 2>&1
tail -4 $O/lr0.05.txt
# lr 0.001 (the metric's): the whole climb
LR=0.001 EPOCHS=40,80,120,160 NSEEDS=8 timeout 2400 python tools/fullepoch_study.py timed_nolds timed_lds strict > $O/lr0.001.txt 2>&1
tail -5 $O/lr0.001.txt
