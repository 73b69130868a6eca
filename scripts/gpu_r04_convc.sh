#!/bin/bash
# centred 3x3 kernel + branch-free byte gather + pointwise convs on the centred GEMM: parity, then configs[2] interleaved
TAG=${1:-r04convc}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_convwave.py tests/test_gpu_fullsize.py tests/test_gpu_gemm256c.py tests/test_gpu_conv_matrix.py tests/test_gpu_operators.py tests/test_gpu_random_shapes.py tests/test_gpu_dwcol.py -q -p no:cacheprovider 2>&1 | tail -n 15 | tee $OUT/pytest.log
timeout 600 python tools/conv33_time.py 3 2>&1 | grep -v amdgpu.ids | tee $OUT/conv33_time.txt
