#!/bin/bash
# round 5: the long-K streaming flavour with the next unit's rows in flight (many rows, few channels) -- parity, then ResNet-50's
# 28x28 512 -> 128 against the generic tile kernel ("gemm_kernel" 1) and the 256-wide kernel (2) on the same box
TAG=${1:-r05longkpf}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_pwstream.py tests/test_gpu_sweep_bench_batch.py tests/test_gpu_residual.py -q -p no:cacheprovider 2>&1 | tail -n 8 | tee $OUT/pytest.log
for shape in "28 28 1 1 1 512 128" "56 56 1 1 1 512 128" "28 28 1 1 1 384 96"; do
  for v in 0 1 2 0 1; do
    timeout 200 python tools/conv_one_time.py $shape 2 $v 2>&1 | grep -v amdgpu.ids | tail -n 1 | sed "s/^/gemm_kernel $v: /" | tee -a $OUT/rows.txt
  done
done
