#!/bin/bash
# r6 parity study, part 2: the edge of the new budget (lr 0.005), lr 0.01 now OUTSIDE it (what auto picks there), and a
# period as 2 / 4 launches at lr 0.05 (a user's triples of a period no longer back to back)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_parity; mkdir -p $O
LR=0.005 EPOCHS=6,9,12,16 NSEEDS=8 timeout 900 python tools/fullepoch_study.py strict auto timed_nolds > $O/lr0.005.txt 2>&1; tail -4 $O/lr0.005.txt
LR=0.01 EPOCHS=3,4,6,8 NSEEDS=8 timeout 900 python tools/fullepoch_study.py reference_nolds ref_lsplit2 auto > $O/lr0.01_outside.txt 2>&1; tail -4 $O/lr0.01_outside.txt
NSEEDS=12 timeout 900 python tools/fullepoch_study.py ref_lsplit2 ref_lsplit4 auto > $O/lr0.05_lsplit.txt 2>&1; tail -4 $O/lr0.05_lsplit.txt
