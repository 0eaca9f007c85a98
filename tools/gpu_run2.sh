python -m pytest tests/test_gpu_fullscale_reference.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -30
python -m pytest tests/test_gpu_bench.py -x -q 2>&1 | tail -6
python tools/e2e_many_seeds.py adam uniform 100 1 strict-own-order,batched,batched-inflight512,batched-inflight256 2>&1 | grep -v amdgpu.ids | grep "epoch 12\|epoch  4" > gpurun_out/vstream_inflight_study.txt; cat gpurun_out/vstream_inflight_study.txt
