#!/bin/bash
# round 5: one iteration on the patch kernel -- parity of the dense ResNet rows, their times, tile flavours A/B (measurement build)
TAG=${1:-r05piter}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_lists_bench_batch.py -q -p no:cacheprovider -x -k "k3s" 2>&1 | tail -n 4 | tee $OUT/pytest.log
timeout 600 python tools/conv_lists_time.py dense3x3 2>&1 | grep -v amdgpu.ids | tee $OUT/conv_lists.txt
export QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so
for SHAPE in "14 14 3 1 1 256 256" "7 7 3 1 1 512 512" "28 28 3 2 1 128 256" "14 14 3 2 1 256 512"; do
  for T in 48 44 48 44; do
    echo -n "tile $T: "
    QNNP_PATCH_TILE=$T timeout 100 python tools/conv_one_time.py $SHAPE 2 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$4,$5,$6,$7, $9, $10}' | tr '\n' ' '
    echo
  done
done | tee $OUT/patch_tiles_ab.txt
