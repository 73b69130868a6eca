#!/bin/bash
# round 6: the alignment-free GEMM with transposed (16-byte / short-dword-short) stores: parity, then the ShuffleNet 1x1 rows
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_gemm128u.py tests/test_gpu_reference_lists.py tests/test_gpu_random_shapes.py tests/test_gpu_pwstream.py tests/test_gpu_conv_matrix.py tests/test_gpu_operators.py -q 2>&1 | tail -8 > gpurun_out/u16st_pytest.log
{
for v in 0 5 24 29; do timeout 100 python tools/conv_one_time.py 7 7 1 1 1 192 1024 3 $v | tail -1; done
for v in 0 5 29; do timeout 100 python tools/conv_one_time.py 7 7 1 1 1 464 1024 3 $v | tail -1; done
for v in 0 5 29; do timeout 100 python tools/conv_one_time.py 56 56 1 1 1 24 60 3 $v | tail -1; done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/u16st_time2.txt
cat gpurun_out/u16st_pytest.log gpurun_out/u16st_time2.txt
