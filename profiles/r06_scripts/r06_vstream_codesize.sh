#!/bin/bash
# r6: k_vstream waits for instruction fetch (HISTORY.md 4.5 / profiles/r04_vstream_direct.md) — does the compiler's own
# size-over-speed setting buy what hand-sharing the optimizer math would?  bpr_vstream.hip rebuilt ON THE BOX with -Os / -O2.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06v; mkdir -p $O
C=revisit-bpr_amd/csrc
FL="-std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-unused-value -ffp-contract=off"
bench() { timeout 300 python bench.py --workload yelp --no-cpu-baseline --steady-epochs 0 --sustained-epochs 3 $2 > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
try:
    j = json.loads(open("$O/$1.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print("%-14s %.1f M triples/s  kernel %.4f ms  frac %.3f" % ("$1", j["value"] / 1e6, r["kernel_ms_avg"], r["frac"]))
except Exception as ex: print("$1 failed", ex)
PY
}
cp $C/bpr_vstream.o $O/bpr_vstream.O3.o; cp revisit-bpr_amd/libbprcore.so $O/libbprcore.O3.so
bench O3 ""; bench O3_bias "--item-bias 1"
for opt in Os O2; do
  /opt/rocm/bin/hipcc -$opt $FL -c $C/bpr_vstream.hip -o $C/bpr_vstream.o 2> $O/build_$opt.log && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o revisit-bpr_amd/libbprcore.so $C/bprcore.o $C/bpr_refresh.o $C/bpr_vstream.o $C/bpr_comm.o $C/bpr_hotlds.o $C/bpr_eval.o -ldl
  ls -la $C/bpr_vstream.o | awk '{print "'$opt' object bytes", $5}'
  bench $opt ""; bench ${opt}_bias "--item-bias 1"
done
cp $O/bpr_vstream.O3.o $C/bpr_vstream.o; cp $O/libbprcore.O3.so revisit-bpr_amd/libbprcore.so; rm -f $O/*.so $O/*.o
