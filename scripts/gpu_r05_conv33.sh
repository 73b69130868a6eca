#!/bin/bash
# round 5: the pipelined unit loop of the weight-stationary 3x3 kernel -- parity, then configs[2] timed through the product
# library and, in the measurement build, one-buffer loop (QNNP_CONV_WS_PIPE=0) against the pipelined one, with cycle stamps
TAG=${1:-r05conv}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_convwave.py tests/test_gpu_fullsize.py tests/test_gpu_conv_matrix.py tests/test_gpu_random_shapes.py -q -p no:cacheprovider -x 2>&1 | tail -n 8 | tee $OUT/pytest.log
echo "== product library"
timeout 300 python tools/conv33_time.py 2 2>&1 | grep -v amdgpu.ids | tee $OUT/conv33_product.txt
export QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so
for R in 1 2; do
  for P in 0 1; do
    echo "== measurement build, QNNP_CONV_WS_PIPE=$P (round $R)"
    QNNP_CONV_WS_PIPE=$P timeout 300 python tools/conv33_time.py 1 2>&1 | grep -v amdgpu.ids | tee -a $OUT/conv33_ab.txt
  done
done
for P in 0 1; do
  echo "== stamps, QNNP_CONV_WS_PIPE=$P"
  QNNP_CONV_WS_PIPE=$P timeout 300 python tools/trace_conv33.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/conv33_stamps_pipe$P.txt
done
