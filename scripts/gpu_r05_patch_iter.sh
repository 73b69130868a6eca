#!/bin/bash
# round 5: one iteration on the patch kernel -- parity of the dense ResNet rows, their times, stamps + ablations of one layer
TAG=${1:-r05piter}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_lists_bench_batch.py -q -p no:cacheprovider -x -k "k3s" 2>&1 | tail -n 4 | tee $OUT/pytest.log
timeout 600 python tools/conv_lists_time.py dense3x3 2>&1 | grep -v amdgpu.ids | tee $OUT/conv_lists.txt
export QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so
timeout 200 python tools/trace_patch.py 14 14 3 1 1 256 256 2>&1 | grep -v amdgpu.ids | tee -a $OUT/patch_stamps.txt
for A in 0 1 2 4 8 15 111 0; do
  echo -n "abl $A: "
  QNNP_PATCH_ABL=$A timeout 100 python tools/conv_one_time.py 14 14 3 1 1 256 256 2 2>&1 | grep -v amdgpu.ids | awk '{print $9, $10}' | tr '\n' ' '
  echo
done | tee $OUT/patch_ablation.txt
