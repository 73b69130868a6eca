#!/bin/bash
# same-box A/B of depthwise kernel variants over the ten MobileNetV2 depthwise layers (sweep indices), batch 128
# usage: gpurun -- bash scripts/gpu_dwab.sh <tag> "<variants>" ["<layers>"]
TAG=${1:-dwab}; VARS=${2:-"0 6"}; LAYERS=${3:-"2 5 8 10 13 15 18 22 24 27"}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for L in $LAYERS; do
  for V in $VARS; do
    timeout 120 python bench.py --layer $L --dw-kernel $V --steps 20 --warmup 3 2>/dev/null | tail -n 1 | sed "s/^/v$V /" | tee -a $OUT/dwab.txt
  done
done
