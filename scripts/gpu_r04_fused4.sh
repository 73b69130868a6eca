#!/bin/bash
# strip kernel cycle stamps, chunk weights in LDS (default) against from L2 ("fused_weights" 2); ablation build
TAG=${1:-r04w2}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export QNNP_GFX950_PRINT_PLAN=1 QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so
timeout 300 python tools/trace_fused.py b7_fused b11_fused b14_fused b16_fused b4_fused 2>&1 | grep -v amdgpu.ids | sort -u | tee $OUT/trace_fused_lds.txt
QNNP_FUSED_WEIGHTS=2 timeout 300 python tools/trace_fused.py b7_fused b11_fused b14_fused b16_fused b4_fused 2>&1 | grep -v amdgpu.ids | sort -u | tee $OUT/trace_fused_l2.txt
