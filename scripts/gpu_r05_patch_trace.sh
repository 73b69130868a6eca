#!/bin/bash
# round 5: cycle and wall-clock stamps of the patch kernel (measurement build) + SQ / LDS counters (product build)
TAG=${1:-r05ptrace}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for SHAPE in "14 14 3 1 1 256 256" "28 28 3 1 1 128 128"; do
  QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so timeout 200 python tools/trace_patch.py $SHAPE 2>&1 | grep -v amdgpu.ids | tee -a $OUT/patch_stamps.txt
done
bash scripts/gpu_pmc_cmd.sh $TAG sq "python tools/conv_one_time.py 14 14 3 1 1 256 256 1" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES | tee $OUT/pmc_sq.txt
bash scripts/gpu_pmc_cmd.sh $TAG lds "python tools/conv_one_time.py 14 14 3 1 1 256 256 1" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM | tee $OUT/pmc_lds.txt
rm -rf $OUT/pmc_sq $OUT/pmc_lds
