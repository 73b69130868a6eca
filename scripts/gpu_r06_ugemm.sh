#!/bin/bash
TAG=${1:-r06p}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_gemm128u.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 25 | tee $OUT/pytest_ugemm.log
timeout 900 python tools/ugemm_time.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ugemm_time.txt
