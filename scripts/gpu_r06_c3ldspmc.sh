#!/bin/bash
# round 6: SQ / LDS / TCC counters of the 7x7 entry layer, register-path (14) against LDS-staged (30) kernel
TAG=${1:-r06u}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for V in 30 14; do
CMD="python tools/conv_one_time.py 224 224 7 2 1 3 64 1 $V"
bash scripts/gpu_pmc_cmd.sh $TAG c3_sq_$V "$CMD" SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU | tee $OUT/pmc_c3_sq_$V.txt
bash scripts/gpu_pmc_cmd.sh $TAG c3_lds_$V "$CMD" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM | tee $OUT/pmc_c3_lds_$V.txt
bash scripts/gpu_pmc_cmd.sh $TAG c3_mfma_$V "$CMD" SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_SMEM | tee $OUT/pmc_c3_mfma_$V.txt
bash scripts/gpu_pmc_cmd.sh $TAG c3_tcc_$V "$CMD" TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_EA0_WRREQ_64B_sum | tee $OUT/pmc_c3_tcc_$V.txt
rm -rf $OUT/pmc_c3_sq_$V $OUT/pmc_c3_lds_$V $OUT/pmc_c3_mfma_$V $OUT/pmc_c3_tcc_$V
done
