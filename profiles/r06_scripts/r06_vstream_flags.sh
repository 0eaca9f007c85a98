#!/bin/bash
# r6: more compiler settings for bpr_vstream.hip (instruction-fetch bound), rebuilt on the box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06w; mkdir -p $O
C=revisit-bpr_amd/csrc
FL="-std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-unused-value -ffp-contract=off"
bench() { timeout 300 python bench.py --workload yelp --no-cpu-baseline --steady-epochs 0 --sustained-epochs 3 $2 > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
try:
    j = json.loads(open("$O/$1.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("%-28s %.1f M triples/s  kernel %.4f ms  frac %.3f" % ("$1", j["value"] / 1e6, r["kernel_ms_avg"], r["frac"]))
except Exception as ex: print("$1 failed", ex)
PY
}
cp $C/bpr_vstream.o $O/keep.o; cp revisit-bpr_amd/libbprcore.so $O/keep.so
bench O2_as_built ""
i=0
for opt in "-O2 -fno-unroll-loops" "-O1" "-O2 -mllvm -amdgpu-sched-strategy=iterative-minreg" "-O2 -mllvm -amdgpu-schedule-metric-bias=0" "-O3 -fno-unroll-loops"; do
  i=$((i+1))
  if /opt/rocm/bin/hipcc $opt $FL -c $C/bpr_vstream.hip -o $C/bpr_vstream.o 2> $O/build_$i.log; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o revisit-bpr_amd/libbprcore.so $C/bprcore.o $C/bpr_refresh.o $C/bpr_vstream.o $C/bpr_comm.o $C/bpr_hotlds.o $C/bpr_eval.o -ldl
    bench "v$i" ""; echo "   v$i = $opt"
  else echo "build failed: $opt"; tail -2 $O/build_$i.log; fi
done
cp $O/keep.o $C/bpr_vstream.o; cp $O/keep.so revisit-bpr_amd/libbprcore.so; rm -f $O/keep.*
