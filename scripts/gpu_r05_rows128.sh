#!/bin/bash
# round 5: ResNet-50's late pointwise layers (few row tiles): the centred 256 x 256 kernel auto picks against 128 x 256 tiles
# ("gemm_kernel" 10, two workgroups per CU) on the same box (measurement build: forced tile shapes exist there only)
TAG=${1:-r05rows128}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for shape in "7 7 1 1 1 2048 512" "7 7 1 1 1 512 2048" "14 14 1 1 1 1024 256" "14 14 1 1 1 1024 512" "28 28 1 1 1 512 256" "28 28 1 1 1 512 128" "28 28 1 2 1 512 1024" "14 14 1 2 1 1024 2048"; do
  for v in 0 10; do
    QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so timeout 200 python tools/conv_one_time.py $shape 2 $v 2>&1 | grep -v amdgpu.ids | tail -n 1 | sed "s/^/variant $v: /" | tee -a $OUT/rows.txt
  done
done
