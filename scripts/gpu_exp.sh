#!/bin/bash
# scratch experiments of the moment (same box): gpurun -- bash scripts/gpu_exp.sh <tag>
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_convstream.py tests/test_gpu_pwstream.py tests/test_gpu_sweep_bench_batch.py tests/test_gpu_dwcol.py tests/test_gpu_requant_corners.py -q -p no:cacheprovider 2>&1 | tail -n 3
bash scripts/gpu_ab_lib.sh $TAG tmp_libs/new3.so tmp_libs/new7.so
