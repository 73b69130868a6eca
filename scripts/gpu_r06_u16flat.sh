#!/bin/bash
# round 6: flat-row stores in the alignment-free GEMM: parity, then the odd-channel 1x1 rows
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_gemm128u.py tests/test_gpu_grouped_dense.py tests/test_gpu_reference_lists.py tests/test_gpu_random_shapes.py tests/test_gpu_pwstream.py tests/test_gpu_operators.py tests/test_gpu_conv_matrix.py -q 2>&1 | tail -6 > gpurun_out/u16flat_pytest.log
timeout 900 python tools/ugemm_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/u16flat_time.txt
cat gpurun_out/u16flat_pytest.log; cat gpurun_out/u16flat_time.txt
