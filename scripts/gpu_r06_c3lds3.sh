#!/bin/bash
# round 6: first-layer kernels after the whole-line stores / LDS staging / centred image: parity over every list shape, timings
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_convc3rows.py tests/test_gpu_requant_packed_tail.py tests/test_gpu_reference_lists.py tests/test_gpu_conv_lists_bench_batch.py tests/test_gpu_random_shapes.py -x -q 2>&1 | tail -5 > gpurun_out/c3lds_pytest.log
{
timeout 120 python tools/conv_one_time.py 224 224 3 1 1 3 64 4 0 | tail -2
timeout 120 python tools/conv_one_time.py 224 224 3 2 1 3 32 4 0 | tail -2
timeout 120 python tools/conv_one_time.py 224 224 3 2 1 3 24 4 0 | tail -2
timeout 120 python tools/conv_one_time.py 224 224 7 2 1 3 64 4 0 | tail -2
timeout 120 python tools/conv_one_time.py 224 224 7 2 1 3 96 4 0 | tail -2
} 2>&1 | grep -v amdgpu.ids > gpurun_out/c3lds_ab.txt
cat gpurun_out/c3lds_pytest.log gpurun_out/c3lds_ab.txt
