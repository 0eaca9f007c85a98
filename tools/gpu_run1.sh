python -m pytest tests/test_gpu_parity.py -x -q -k "async_cut or sync_refresh_right_behind" 2>&1 | tail -5
python -m pytest tests/test_gpu_two_tier.py -x -q 2>&1 | tail -5
mkdir -p gpurun_out/nosort
for args in "" "--refresh-cus 0" "--refresh-cus 32" "--refresh-cus 16"; do
  BPR_LIB_PATH=$PWD/tools/ubench/libbprcore_nosort.so python bench.py --steps 96 --warmup 8 --no-cpu-baseline --sustained-epochs 0 $args > gpurun_out/nosort/b.json 2>gpurun_out/nosort/b.err
  python - "$args" <<PY
import json,sys
j=json.loads(open("gpurun_out/nosort/b.json").read().strip().splitlines()[-1])
print("NOSORT %-20s value %.1f M  ms/step %.4f  kernel %.4f ms" % (sys.argv[1], j["value"]/1e6, j["ms_per_step"], j["roofline"]["kernel_ms_avg"]))
PY
done
for args in "" "--refresh-cus 32" "--refresh-cus 48"; do
  python bench.py --steps 96 --warmup 8 --no-cpu-baseline --sustained-epochs 0 $args > gpurun_out/nosort/b.json 2>gpurun_out/nosort/b.err
  python - "$args" <<PY
import json,sys
j=json.loads(open("gpurun_out/nosort/b.json").read().strip().splitlines()[-1])
print("SORT   %-20s value %.1f M  ms/step %.4f  kernel %.4f ms" % (sys.argv[1], j["value"]/1e6, j["ms_per_step"], j["roofline"]["kernel_ms_avg"]))
PY
done
