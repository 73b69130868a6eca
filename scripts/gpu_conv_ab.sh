#!/bin/bash
# 3x3 convolution (configs[2]) kernels: parity tests, then alternating timings of "gemm_kernel" variants
#   gpu_conv_ab.sh <tag> "<variants>"      e.g. "0 12"
TAG=${1:-convab}; VARS=${2:-"0 12"}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== parity"
timeout 600 python -m pytest tests/test_gpu_convwave.py tests/test_gpu_fullsize.py -m gpu -q -x -k "wave or c3" -p no:cacheprovider 2>&1 | tail -n 8 | tee $OUT/pytest.log
echo "== timing (alternating)"
for round in 1 2 3; do
  for V in $VARS; do
    timeout 120 python bench.py --layer 99 --steps 30 --warmup 5 --gemm-kernel $V 2>/dev/null | tail -n 1 | tee -a $OUT/conv3x3_v$V.txt
  done
done
