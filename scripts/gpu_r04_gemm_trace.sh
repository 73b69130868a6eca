#!/bin/bash
# round 4: cycle stamps of the centred GEMM (ablation build), the ablations in cycles AND time, the spread-reads A/B
TAG=${1:-r04c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so
for A in 0 64 1 8 16 2 32 27; do
  GEMM_KERNEL=20 QNNP_GFX950_ABLATE=$A timeout 120 python tools/trace_gemm_c.py 2>&1 | grep kernel | tee -a $OUT/trace_gemm_c.txt
done
for V in 22 21; do
  GEMM_KERNEL=$V QNNP_GFX950_ABLATE=0 timeout 120 python tools/trace_gemm_c.py 2>&1 | grep kernel | tee -a $OUT/trace_gemm_c.txt
done
unset QNNP_GFX950_LIBRARY
echo "== A/B spread reads"
timeout 300 python tools/gemm_ab.py --variants 20,25,15 --rounds 7 2>&1 | tee $OUT/gemm_ab_spread.txt | tail -n 4
