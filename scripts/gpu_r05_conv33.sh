#!/bin/bash
# round 5: the weight-stationary 3x3 kernel -- parity, configs[2] timed through the product library, cycle stamps (measurement build)
TAG=${1:-r05conv}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_convwave.py tests/test_gpu_fullsize.py tests/test_gpu_conv_matrix.py tests/test_gpu_random_shapes.py -q -p no:cacheprovider -x 2>&1 | tail -n 8 | tee $OUT/pytest.log
echo "== product library"
timeout 300 python tools/conv33_time.py 3 2>&1 | grep -v amdgpu.ids | tee $OUT/conv33_product.txt
export QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so
echo "== stamps"
timeout 300 python tools/trace_conv33.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/conv33_stamps.txt
