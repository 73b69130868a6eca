#!/bin/bash
# round 5: what the patch kernel's launch is made of -- one phase removed at a time (measurement build, QNNP_PATCH_ABL), same box
TAG=${1:-r05pabl}; SHAPE=${2:-"14 14 3 1 1 256 256"}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so
for A in 0 16 1 2 3 4 8 32 64 12 15 47 111 0; do
  echo -n "abl $A: "
  QNNP_PATCH_ABL=$A timeout 100 python tools/conv_one_time.py $SHAPE 2 2>&1 | grep -v amdgpu.ids | awk '{print $9, $10}' | tr '\n' ' '
  echo
done | tee $OUT/patch_ablation.txt
