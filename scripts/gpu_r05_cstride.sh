#!/bin/bash
# round 5: strided 1x1 convolutions with K >= 512 on the zero-point-centred 256 x 256 GEMM (table rows) -- parity, then the two
# ResNet-50 rows against the measurement build's previous dispatch (offset-table flavour, "gemm_kernel" 2) on the same box
TAG=${1:-r05cstride}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_gemm256c.py tests/test_gpu_conv_lists_bench_batch.py tests/test_gpu_conv_matrix.py tests/test_gpu_operators.py -q -p no:cacheprovider 2>&1 | tail -n 8 | tee $OUT/pytest.log
for shape in "28 28 1 2 1 512 1024" "14 14 1 2 1 1024 2048" "56 56 1 2 1 512 512"; do
  for v in 0 2 0 2; do
    timeout 200 python tools/conv_one_time.py $shape 2 $v 2>&1 | grep -v amdgpu.ids | tail -n 1 | sed "s/^/gemm_kernel $v: /" | tee -a $OUT/rows.txt
  done
done
