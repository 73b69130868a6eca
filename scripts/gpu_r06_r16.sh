#!/bin/bash
TAG=${1:-r06h}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_gemm256c.py tests/test_gpu_requant_packed_tail.py tests/test_gpu_fullsize.py -m gpu -q --maxfail=10 -p no:cacheprovider 2>&1 | tail -n 25 | tee $OUT/pytest_r16.log
timeout 600 python tools/gemm_ab.py --variants 15,28 --kzp 126 --rounds 7 2>&1 | grep -v amdgpu.ids | tee $OUT/gemm_ab_kzp126.txt
