#!/bin/bash
# PMC evidence for the kernels added in round 3: SQ (issue / wait) and TCC (fabric traffic) passes, one command each
TAG=${1:-pmcnew}; export TMPDIR=/tmp
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
TCC="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum"
run() { # name command
  bash scripts/gpu_pmc_cmd.sh $TAG $1_sq "$2" $SQ 2>&1 | grep -E "q8_|no counter" | grep -v "probe" | tee gpurun_out/$TAG/pmc_$1_sq.txt
  bash scripts/gpu_pmc_cmd.sh $TAG $1_tcc "$2" $TCC 2>&1 | grep -E "q8_|no counter" | grep -v "probe" | tee gpurun_out/$TAG/pmc_$1_tcc.txt
}
mkdir -p gpurun_out/$TAG
run c3rows "python bench.py --layer 1 --steps 20 --warmup 3"
run convws "python bench.py --layer 99 --steps 20 --warmup 3"
run dw8 "python bench.py --layer 8 --steps 20 --warmup 3"
DW5_ONLY=28x28 DW5_ROUNDS=1 run dw5 "python tools/dw5_time.py 128 0"
run deconv "python tools/next_rows_time.py 128 0"
