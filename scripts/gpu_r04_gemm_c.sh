#!/bin/bash
# round 4, step 1: the zero-point-centred GEMM flavours ("gemm_kernel" 20..23): parity, interleaved A/B against the lean
# flavour (15), one SQ counter pass each for 15 and the winner candidates
TAG=${1:-r04a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== parity (centred kernel)"
timeout 900 python -m pytest tests/test_gpu_gemm256c.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 12 | tee $OUT/pytest_gemm256c.log
echo "== parity (4096^3, all structures; names of the auto picks)"
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_gemm256.py -m gpu -q -p no:cacheprovider -k "c2_q8gemm or auto_takes" 2>&1 | tail -n 12 | tee $OUT/pytest_fullsize_gemm.log
echo "== A/B"
timeout 400 python tools/gemm_ab.py --variants 15,20,21,22,23 --rounds 7 2>&1 | tee $OUT/gemm_ab.txt | tail -n 8
SQA="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
for V in 15 21 23; do
  bash scripts/gpu_pmc_cmd.sh $TAG gemm${V}_sq "python tools/gemm_ab.py --only $V" $SQA | tee $OUT/pmc_gemm${V}_sq.txt
done
