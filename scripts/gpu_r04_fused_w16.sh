#!/bin/bash
# fused strip kernel: 8 waves x <= 256 registers (two tiles per round, prefetch) against 16 waves x <= 128 registers
TAG=${1:-r04l}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for L in libqnnpack_gfx950.so libqnnpack_gfx950_w16.so; do
  echo "== $L"
  QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/$L timeout 900 python -m pytest tests/test_gpu_fused.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 3
  QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/$L timeout 300 python tools/network_profile.py 128 fuse 2>&1 | grep -v amdgpu.ids | tee $OUT/network_per_operator_fused_$L.txt | head -20
done
