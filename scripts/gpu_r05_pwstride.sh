#!/bin/bash
# round 5: strided 1x1 convolutions on the staged pointwise streaming kernel (a table row per output pixel) -- parity on the
# ResNet lists and the operator sweeps, then the four ResNet rows against the previous build (OLD) on the same box
TAG=${1:-r05pws}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
OLD=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so
timeout 1500 python -m pytest tests/test_gpu_pwstream.py tests/test_gpu_conv_lists_bench_batch.py tests/test_gpu_conv_matrix.py tests/test_gpu_operators.py tests/test_gpu_random_shapes.py tests/test_gpu_fullsize.py -q -p no:cacheprovider 2>&1 | tail -n 12 | tee $OUT/pytest.log
for shape in "56 56 1 2 1 256 512" "56 56 1 2 1 64 128" "28 28 1 2 1 128 256" "14 14 1 2 1 256 512"; do
  echo "== $shape product" | tee -a $OUT/rows.txt
  timeout 200 python tools/conv_one_time.py $shape 2 2>&1 | grep -v amdgpu.ids | tee -a $OUT/rows.txt
  echo "== $shape previous build" | tee -a $OUT/rows.txt
  QNNP_GFX950_LIBRARY=$OLD timeout 200 python tools/conv_one_time.py $shape 2 2>&1 | grep -v amdgpu.ids | tee -a $OUT/rows.txt
done
