#!/bin/bash
TAG=${1:-r06g}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_convwave.py tests/test_gpu_random_shapes.py tests/test_gpu_fullsize.py tests/test_gpu_conv_lists_bench_batch.py -m gpu -q --maxfail=10 -p no:cacheprovider 2>&1 | tail -n 25 | tee $OUT/pytest_convwave.log
timeout 600 python tools/conv33_ab.py 6 2>&1 | grep -v amdgpu.ids | tee $OUT/conv33_ab.txt
