#!/bin/bash
# usage: gpurun -- bash scripts/gpu_tests_then_ab.sh <tag> <a.so> <b.so>   -- sweep dispatch table refresh, GPU tier, then a same-box A/B of two library builds
TAG=${1:-r02l}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
QNNP_WRITE_SWEEP_KERNELS=1 timeout 600 python -m pytest tests/test_gpu_sweep_bench_batch.py -q -p no:cacheprovider 2>&1 | tail -n 3
cp tests/golden/sweep_kernels.json $OUT/
timeout 1200 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider 2>&1 | tail -n 60 > $OUT/pytest_gpu.log
tail -n 12 $OUT/pytest_gpu.log
bash scripts/gpu_ab_lib.sh $TAG $2 $3
