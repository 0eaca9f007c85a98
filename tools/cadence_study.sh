#!/bin/bash
# Full-scale (ML-20M-shaped, d=128) multi-rank cadence study on ONE GPU over gloo: nDCG@100 / Recall@20
# of 1 / 2 / 4 ranks with a full refresh period per RANK and chunk (bench.py --cadence rank) and with
# the period divided by the ranks (--cadence job).   tools/cadence_study.sh > profiles/r03_cadence_study.txt
cd "$(dirname "$0")/.."
export BPR_DIST_BACKEND=gloo BPR_EPOCHS=4
run() { # world cadence
  if [ "$1" = 1 ]; then BPR_CADENCE=$2 python tools/parity_multi.py adaptive 1,2,3 stream full
  else BPR_CADENCE=$2 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 2966$1 tools/parity_multi.py adaptive 1,2,3 stream full; fi 2>/dev/null | python -c "
import sys, json, numpy as np
runs = [json.loads(l) for l in sys.stdin if l.startswith('{')]
nd = np.array([r['ndcg@100'] for r in runs]); rc = np.array([r['recall@20'] for r in runs])
print('world %s cadence %-4s seeds %d  nDCG@100 per epoch %s (last +- %.4f)  Recall@20 last %.4f' % (sys.argv[1], sys.argv[2], len(runs), np.round(nd.mean(0), 4).tolist(), nd[:, -1].std(ddof=1), rc[:, -1].mean()))
" $1 $2; }
run 1 job
for w in 2 4; do run $w rank; run $w job; done
