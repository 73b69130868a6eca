#!/bin/bash
# HBM-side traffic (TCC) and SQ counters of the depthwise kernels, round 6 (the last TCC pass on a depthwise layer was round 3's)
TAG=${1:-r06j}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
bash scripts/gpu_pmc_cmd.sh $TAG dw8_tcc "python bench.py --layer 8 --steps 30 --warmup 3" TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_EA0_WRREQ_64B_sum | tee $OUT/pmc_dw8_tcc.txt
bash scripts/gpu_pmc_cmd.sh $TAG dw5_tcc "python bench.py --layer 5 --steps 30 --warmup 3" TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_EA0_WRREQ_64B_sum | tee $OUT/pmc_dw5_tcc.txt
bash scripts/gpu_pmc_cmd.sh $TAG dw5x5_tcc "DW5_ONLY=28x28x240 DW5_ROUNDS=3 python tools/dw5_time.py" TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_EA0_WRREQ_64B_sum | tee $OUT/pmc_dw5x5_tcc.txt
rm -rf $OUT/pmc_dw8_tcc $OUT/pmc_dw5_tcc $OUT/pmc_dw5x5_tcc
timeout 300 python -m pytest tests/test_gpu_gemm256.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
