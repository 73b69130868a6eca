#!/bin/bash
# round 3, first GPU call: GPU tier (new discriminating batch-128 tests, GEMM structures 10 / 11), interleaved GEMM A/B,
# the PMC passes the round-2 review found missing (3x3 convolution: none existed; GEMM: round-1 counters only), bench line.
TAG=${1:-r03a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider 2>&1 | tail -n 60 > $OUT/pytest_gpu.log; tail -n 25 $OUT/pytest_gpu.log
echo "== GEMM structures, interleaved"
timeout 300 python tools/gemm_ab.py --variants 0,10,11,4 --rounds 5 2>&1 | tee $OUT/gemm_ab.txt | tail -n 6
SQA="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU"
SQB="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVES"
TCC="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_EA0_WRREQ_64B_sum"
echo "== PMC: 3x3 convolution (configs[2])"
CONV="python bench.py --layer 99 --steps 20 --warmup 3"
bash scripts/gpu_pmc_cmd.sh $TAG conv3x3_sq_a "$CONV" $SQA | tee $OUT/pmc_conv3x3_a.txt
bash scripts/gpu_pmc_cmd.sh $TAG conv3x3_sq_b "$CONV" $SQB | tee $OUT/pmc_conv3x3_b.txt
bash scripts/gpu_pmc_cmd.sh $TAG conv3x3_tcc "$CONV" $TCC | tee $OUT/pmc_conv3x3_c.txt
echo "== PMC: GEMM structures"
for V in 0 10 11; do
  bash scripts/gpu_pmc_cmd.sh $TAG gemm${V}_sq_a "python tools/gemm_ab.py --only $V" $SQA | tee $OUT/pmc_gemm${V}_a.txt
  bash scripts/gpu_pmc_cmd.sh $TAG gemm${V}_sq_b "python tools/gemm_ab.py --only $V" $SQB | tee $OUT/pmc_gemm${V}_b.txt
done
for V in 0 10; do
  bash scripts/gpu_pmc_cmd.sh $TAG gemm${V}_tcc "python tools/gemm_ab.py --only $V" $TCC | tee $OUT/pmc_gemm${V}_c.txt
done
echo "== bench"
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -n 2 | tee $OUT/bench.json | cut -c1-1500
