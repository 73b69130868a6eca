#!/bin/bash
# round 5: cycle and wall-clock stamps of the patch kernel (measurement build), full and with the loop's content removed
TAG=${1:-r05ptrace}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so
for A in 0 111; do
  echo "== QNNP_PATCH_ABL=$A"
  QNNP_PATCH_ABL=$A timeout 200 python tools/trace_patch.py ${2:-14 14 3 1 1 256 256} 2>&1 | grep -v amdgpu.ids | tee -a $OUT/patch_stamps.txt
done
