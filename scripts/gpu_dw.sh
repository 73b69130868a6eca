#!/bin/bash
# DW kernel tuning: tests + per-layer timings under different LDS budgets
export TMPDIR=/tmp; mkdir -p gpurun_out/$1
timeout 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_fullsize.py -q -m gpu -k "depthwise or dw or golden" --maxfail=10 -p no:cacheprovider 2>&1 | tail -n 8
for kb in ${2:-32 48 64 24}; do
  echo "== LDS budget $kb KB"
  for L in 2 5 8 10 13 15 18 22 24 27; do
    QNNP_GFX950_DW_LDS_KB=$kb timeout 120 python bench.py --layer $L --steps 20 --warmup 3 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['layer'], d['shape'], d['ms'], d['gbs'])"
  done
done | tee gpurun_out/$1/dw.txt
