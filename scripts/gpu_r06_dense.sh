#!/bin/bash
# round 6: grouped 1x1 as one dense GEMM -- parity, then the ShuffleNet 1x1 rows under auto
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_grouped_dense.py tests/test_gpu_gemm128u.py tests/test_gpu_reference_lists.py tests/test_gpu_conv_lists_bench_batch.py tests/test_gpu_operators.py -q 2>&1 | tail -8 > gpurun_out/dense_pytest.log
timeout 900 python tools/ugemm_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/dense_time.txt
cat gpurun_out/dense_pytest.log; grep -E "^\[28|sums" gpurun_out/dense_time.txt
