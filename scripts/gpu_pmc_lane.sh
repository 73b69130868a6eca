#!/bin/bash
# SQ counters of the lane-form kernels: the weight-stationary 3x3 convolution (compare pmc_convws_sq_r03x.txt) and
# pointwise sweep layers 7 and 21 (staged kernel)
TAG=${1:-r03lanepmc}; OUT=gpurun_out/$TAG; mkdir -p $OUT
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
bash scripts/gpu_pmc_layer2.sh $TAG convws_sq 99 $SQ | tee $OUT/pmc_convws_sq.txt
bash scripts/gpu_pmc_layer2.sh $TAG pw7_sq 7 $SQ | tee $OUT/pmc_pw7_sq.txt
bash scripts/gpu_pmc_layer2.sh $TAG pw21_sq 21 $SQ | tee $OUT/pmc_pw21_sq.txt
