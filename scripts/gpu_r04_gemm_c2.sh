#!/bin/bash
# round 4, step 2: the simplified centred GEMM (ring 4, bias line by LDS-DMA, clamp classes, weight reads one per MFMA):
# parity, A/B against lean and the burst / nt / sc1 structures, cycle stamps
TAG=${1:-r04d}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== parity (centred kernel)"
timeout 900 python -m pytest tests/test_gpu_gemm256c.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 12 | tee $OUT/pytest_gemm256c.log
echo "== parity (4096^3)"
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_gemm256.py -m gpu -q -p no:cacheprovider -k "c2_q8gemm or auto_takes" 2>&1 | tail -n 12 | tee $OUT/pytest_fullsize_gemm.log
echo "== A/B"
timeout 400 python tools/gemm_ab.py --variants 15,20,21,22,23 --rounds 7 2>&1 | tee $OUT/gemm_ab.txt | tail -n 8
if [ -f qnnpack_amd/libqnnpack_gfx950_abl.so ]; then
  export QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so
  GEMM_KERNEL=20 QNNP_GFX950_ABLATE=0 timeout 120 python tools/trace_gemm_c.py 2>&1 | grep kernel | tee -a $OUT/trace_gemm_c.txt
  unset QNNP_GFX950_LIBRARY
fi
