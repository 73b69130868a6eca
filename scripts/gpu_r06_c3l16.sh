#!/bin/bash
# round 6: the LDS-staged 16-byte-slot first-layer kernel: parity, then the first layers of the lists
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_convc3rows.py tests/test_gpu_requant_packed_tail.py tests/test_gpu_random_shapes.py tests/test_gpu_reference_lists.py tests/test_gpu_conv_lists_bench_batch.py tests/test_gpu_sweep_bench_batch.py -q 2>&1 | tail -8 > gpurun_out/c3l16_pytest.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -k first 2>&1 | tail -3 >> gpurun_out/c3l16_pytest.log
cat gpurun_out/c3l16_pytest.log
{
for shape in "224 224 3 2 1 3 32" "224 224 3 2 1 3 24" "224 224 3 2 1 3 64" "224 224 3 1 1 3 64"; do
  for v in 14 30 14 30; do timeout 100 python tools/conv_one_time.py $shape 4 $v 2>&1 | tail -1; done
done
} | grep -v amdgpu.ids > gpurun_out/c3l16_time.txt
cat gpurun_out/c3l16_time.txt
