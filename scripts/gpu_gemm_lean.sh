#!/bin/bash
# the lean GEMM flavour ("gemm_kernel" = 15): parity, then an interleaved A/B against the shipped kernel
TAG=${1:-r03lean}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== parity"
timeout 600 python -m pytest tests/test_gpu_gemm256.py tests/test_gpu_fullsize.py -m gpu -q -k "lean or 4096" -p no:cacheprovider 2>&1 | tail -n 15 | tee $OUT/pytest_lean.log
echo "== A/B"
timeout 300 python tools/gemm_ab.py --variants ${2:-0,15,2} --rounds 7 2>&1 | tee $OUT/gemm_ab.txt | tail -n 8
