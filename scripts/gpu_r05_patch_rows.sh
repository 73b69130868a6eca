#!/bin/bash
# round 5: the patch kernel's tile height on the HBM-bound stride-2 shapes -- fewer rows per tile = smaller patch = two
# workgroups per CU (one's patch load under the other's multiplies) against more halo rows; measurement build knobs
TAG=${1:-r05prows}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ABL=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so
for SHAPE in "56 56 3 2 1 128 128" "56 56 3 2 1 64 128" "28 28 3 2 1 128 256" "28 28 3 1 1 128 128"; do
  for cfg in "0 0" "44 0" "44 3" "44 2" "84 4" "84 3" "84 2" "48 2"; do
    set -- $cfg
    echo "== $SHAPE tile=$1 rows<=$2" | tee -a $OUT/rows.txt
    QNNP_GFX950_LIBRARY=$ABL QNNP_PATCH_TILE=$1 QNNP_PATCH_ROWS=$2 timeout 100 python tools/conv_one_time.py $SHAPE 2 2>&1 | grep -v amdgpu.ids | tail -n 1 | tee -a $OUT/rows.txt
  done
done
