#!/bin/bash
# three PMC passes (SQ a, SQ b, TCC) over one command:  gpu_pmc3.sh <tag> <name> "<command>"
TAG=$1; NAME=$2; CMD=$3
SQA="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU"
SQB="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVES"
TCC="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_EA0_WRREQ_64B_sum"
bash scripts/gpu_pmc_cmd.sh $TAG ${NAME}_sq_a "$CMD" $SQA
bash scripts/gpu_pmc_cmd.sh $TAG ${NAME}_sq_b "$CMD" $SQB
[ "$4" = "notcc" ] || bash scripts/gpu_pmc_cmd.sh $TAG ${NAME}_tcc "$CMD" $TCC
