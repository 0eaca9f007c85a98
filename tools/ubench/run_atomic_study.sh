#!/bin/bash
# The atomic-unit study behind DESIGN §4.1, in one GPU-box call:  bash tools/ubench/run_atomic_study.sh r02
# (binaries are built in the build container: hipcc -O3 --offload-arch=gfx950 X.hip -o X)
T=${1:-r02}
R=/root/repo; O=$R/gpurun_out; U=$R/tools/ubench
{
  echo "# atomic_bench (tools/ubench/atomic_bench.hip), MI355X, $(date -u +%F)"
  for hot in 2 1 0; do $U/atomic_bench 199168 136678 20109 $hot; echo; done
  echo "# gs_bench (tools/ubench/gs_bench.hip)"
  $U/gs_bench
} > $O/ubench_atomic_$T.txt 2>&1
tail -50 $O/ubench_atomic_$T.txt
