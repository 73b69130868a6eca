#!/bin/bash
# 5x5 depthwise column walk (kernel H): parity, then the bench's three 5x5 shapes against the LDS-tiled kernel
TAG=${1:-dw5}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dwcol5.py tests/test_gpu_dwcol.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 15 | tee $OUT/pytest.log
timeout 300 python tools/dw5_time.py 128 2 0 2>&1 | grep -v amdgpu.ids | tee $OUT/dw5_ab.txt
