#!/bin/bash
TAG=${1:-r06o}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for V in 7 0; do
bash scripts/gpu_pmc_cmd.sh $TAG dwm_sq_$V "python bench.py --layer 8 --dw-kernel $V --steps 30 --warmup 3" SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU | tee $OUT/pmc_dwm_sq_$V.txt
bash scripts/gpu_pmc_cmd.sh $TAG dwm_lds_$V "python bench.py --layer 8 --dw-kernel $V --steps 30 --warmup 3" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM | tee $OUT/pmc_dwm_lds_$V.txt
bash scripts/gpu_pmc_cmd.sh $TAG dwm_tcc_$V "python bench.py --layer 8 --dw-kernel $V --steps 30 --warmup 3" TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum | tee $OUT/pmc_dwm_tcc_$V.txt
rm -rf $OUT/pmc_dwm_sq_$V $OUT/pmc_dwm_lds_$V $OUT/pmc_dwm_tcc_$V
done
