#!/bin/bash
# lean GEMM flavour: the GEMM test files in full, counters (two SQ passes + TCC) for it and for the general flavour
TAG=${1:-r03lean3}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gemm256.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider 2>&1 | tail -n 8 | tee $OUT/pytest_gemm.log
SQA="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU"
SQB="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_WAVES"
TCC="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_EA0_WRREQ_64B_sum"
for V in 0 2; do
  bash scripts/gpu_pmc_cmd.sh $TAG gemm${V}_sq_a "python tools/gemm_ab.py --only $V" $SQA | tee $OUT/pmc_gemm${V}_a.txt
  bash scripts/gpu_pmc_cmd.sh $TAG gemm${V}_sq_b "python tools/gemm_ab.py --only $V" $SQB | tee $OUT/pmc_gemm${V}_b.txt
done
bash scripts/gpu_pmc_cmd.sh $TAG gemm0_tcc "python tools/gemm_ab.py --only 0" $TCC | tee $OUT/pmc_gemm0_c.txt
timeout 200 python tools/gemm_ab.py --variants 0,2 --rounds 7 2>&1 | tee $OUT/gemm_ab.txt | tail -n 3
