#!/bin/bash
# Round 6: the 16x16x64 flavour of the centred GEMM -- parity, then the interleaved A/B against the 32x32x32 one.
TAG=${1:-r06b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest gemm256c (both flavours)"
timeout 1200 python -m pytest tests/test_gpu_gemm256c.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 15 | tee $OUT/pytest_gemm256c.log
echo "== A/B"
timeout 600 python tools/gemm_ab.py --variants 20,23 --rounds 7 2>&1 | grep -v amdgpu.ids | tee $OUT/gemm_ab_c_vs_c16.txt
