#!/bin/bash
# same-box A/B of two library builds on single sweep layers: gpu_ab_layers_lib.sh <tag> <a.so> <b.so> "<layers>"
TAG=${1:-ablay}; A=$2; B=$3; LAYERS=${4:-"6 9 11 14 19 23 25 28 29 30"}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for round in 1 2 3; do
  for L in $LAYERS; do
    for LIB in $A $B; do
      QNNP_GFX950_LIBRARY=$PWD/$LIB timeout 120 python bench.py --layer $L --steps 30 --warmup 5 2>/dev/null | tail -n 1 | sed "s|^|$(basename $LIB .so) |" | tee -a $OUT/layers.txt
    done
  done
done
