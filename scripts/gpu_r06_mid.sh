#!/bin/bash
TAG=${1:-r06d}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest gemm128x"
timeout 1200 python -m pytest tests/test_gpu_gemm128x.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 25 | tee $OUT/pytest_gemm128x.log
echo "== timing"
timeout 900 python tools/mid_gemm_time.py 2>&1 | grep -v amdgpu.ids | tee $OUT/mid_gemm_time.txt
