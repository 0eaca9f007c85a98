#!/bin/bash
# usage: bash tools/repeat_two_rank.sh [tree root] [runs]: the 2-rank stream-lag parity run (tests/test_gpu_multirank_parity.py) over and
# over, counting hangs and crashes.  r5: 7 of 33 runs hung or aborted ("std::system_error: Invalid argument" inside the HIP runtime) because
# bpr_ctx_destroy synchronised a CU-masked stream the garbage collector had already destroyed; 30 of 30 pass since it waits for the device.
export BPR_DIST_BACKEND=gloo
root=${1:-.}; runs=${2:-8}
ok=0; bad=0
for i in $(seq 1 $runs); do
  timeout 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29700+i)) $root/tools/parity_multi.py adaptive 1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30 stream-lag > gpurun_out/dbg_multi.out 2> gpurun_out/dbg_multi.err
  rc=$?
  if [ $rc -eq 0 ]; then ok=$((ok+1)); else bad=$((bad+1)); echo "  run $i rc=$rc after $(grep -c '^{' gpurun_out/dbg_multi.out) seeds"; cp gpurun_out/dbg_multi.err gpurun_out/dbg_multi_fail_$i.err; grep -v "^W\|amdgpu.ids\|^\[W\|OMP_NUM" gpurun_out/dbg_multi.err | grep -v "torch/distributed/\|runpy.py" | head -25; fi
done
echo "$root: ok=$ok bad=$bad"
