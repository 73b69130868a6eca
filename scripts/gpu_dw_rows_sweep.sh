#!/bin/bash
# rows-per-segment sweep of the depthwise column walks (QNNP_GFX950_DW_COL_ROWS, 0 = the plan's own choice)
TAG=${1:-dwrows}; ROWS=${2:-"0 14 28 56 112"}; LAYERS=${3:-"2 5 8 10 13 15 18 22 24 27"}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for L in $LAYERS; do
  for R in $ROWS; do
    QNNP_GFX950_DW_COL_ROWS=$R timeout 120 python bench.py --layer $L --steps 30 --warmup 5 2>/dev/null | tail -n 1 | sed "s|^|rows $R |" | tee -a $OUT/rows.txt
  done
done
