#!/bin/bash
# round 5: channel blocks per workgroup column of the staged pointwise kernel (QNNP_PW_NBP, measurement build) on ResNet-50's
# wide pointwise rows: fewer, wider columns re-read the activations less often
TAG=${1:-r05pwnbp}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ABL=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so
for SHAPE in "14 14 1 1 1 256 1024" "28 28 1 1 1 128 512" "56 56 1 1 1 64 256" "56 56 1 2 1 256 512" "14 14 1 1 1 96 576" "7 7 1 1 1 160 960"; do
  for nbp in 0 2 4 8 16 32; do
    echo -n "$SHAPE nbp=$nbp: " | tee -a $OUT/rows.txt
    QNNP_GFX950_LIBRARY=$ABL QNNP_PW_NBP=$nbp timeout 100 python tools/conv_one_time.py $SHAPE 2 2>&1 | grep -v amdgpu.ids | tail -n 1 | cut -c30-110 | tee -a $OUT/rows.txt
  done
done
