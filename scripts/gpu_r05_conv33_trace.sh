#!/bin/bash
# round 5: where the 3x3 kernel's launch goes -- prologue and unit-loop stamps, wall clock at entry / exit (measurement build)
TAG=${1:-r05trace}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so
for P in 0 1; do
  echo "== QNNP_CONV_WS_PIPE=$P"
  QNNP_CONV_WS_PIPE=$P timeout 300 python tools/trace_conv33.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/conv33_prologue_stamps_pipe$P.txt
done
