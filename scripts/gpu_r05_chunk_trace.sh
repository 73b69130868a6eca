#!/bin/bash
# round 5: cycle / wall-clock stamps of the channel-chunked patch flavour (measurement build), beside its tile alternatives
TAG=${1:-r05chunktr}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ABL=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so
for SHAPE in "14 14 3 2 1 512 512" "28 28 3 2 1 256 256"; do
  QNNP_GFX950_LIBRARY=$ABL timeout 200 python tools/trace_patch.py $SHAPE 2>&1 | grep -v amdgpu.ids | tee -a $OUT/stamps.txt
  for tile in 48 44 84; do
    echo "== $SHAPE QNNP_PATCH_TILE=$tile" | tee -a $OUT/stamps.txt
    QNNP_GFX950_LIBRARY=$ABL QNNP_PATCH_TILE=$tile timeout 200 python tools/conv_one_time.py $SHAPE 2 2>&1 | grep -v amdgpu.ids | tail -n 1 | tee -a $OUT/stamps.txt
  done
done
