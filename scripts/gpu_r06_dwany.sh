#!/bin/bash
# round 6: the sliding-window depthwise kernel on unaligned dwords: parity, then ShuffleNet v2's odd-channel layers
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_dwdirect4.py tests/test_gpu_grouped_dense.py tests/test_gpu_reference_lists.py tests/test_gpu_gemm128u.py tests/test_gpu_pwstream.py -q 2>&1 | tail -8 > gpurun_out/dwany_pytest.log
timeout 600 python tools/dw_any_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/dwany_time.txt
cat gpurun_out/dwany_pytest.log gpurun_out/dwany_time.txt
