#!/bin/bash
# measurement only: VALU occupancy counters of the sweep layers given in $1 (one rocprofv3 --pmc pass per layer)
TAG=${2:-valu}
for layer in $1; do
  echo "== layer $layer"
  bash scripts/gpu_pmc_layer2.sh $TAG valu_l$layer $layer SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES
done
