#!/bin/bash
# round 5: counters of the patch kernel on one ResNet layer (default 14x14 3x3 256 -> 256)
TAG=${1:-r05pmc}; SHAPE=${2:-"14 14 3 1 1 256 256"}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 200 python tools/conv_one_time.py $SHAPE 3 2>&1 | grep -v amdgpu.ids | tee $OUT/time.txt
bash scripts/gpu_pmc_cmd.sh $TAG sq "python tools/conv_one_time.py $SHAPE 1" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES | tee $OUT/pmc_sq.txt
bash scripts/gpu_pmc_cmd.sh $TAG lds "python tools/conv_one_time.py $SHAPE 1" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM | tee $OUT/pmc_lds.txt
rm -rf $OUT/pmc_sq $OUT/pmc_lds
