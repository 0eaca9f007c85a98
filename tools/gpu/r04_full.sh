#!/bin/bash
# r4: the whole GPU suite, then everything profiles/ is built from, then the other shapes
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out; mkdir -p $O
( time timeout 3000 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu_r04.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu_r04.log
tail -8 $O/pytest_gpu_r04.log
python __graft_entry__.py smoke > $O/smoke_r04.log 2>&1; tail -2 $O/smoke_r04.log
bash tools/profile_round.sh r04 > $O/profile_round_r04.log 2>&1; tail -12 $O/profile_round_r04.log | cut -c1-300
cd $R
bash tools/bench_shapes.sh > $O/shapes_r04.txt 2>&1; cat $O/shapes_r04.txt | cut -c1-200
BPR_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29688 bench.py --gpus 2 --steps 40 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_r04_2ranks_gloo.json; cut -c1-600 $O/bench_r04_2ranks_gloo.json
