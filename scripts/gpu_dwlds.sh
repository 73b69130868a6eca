#!/bin/bash
# measurement only: LDS depthwise kernel per sweep layer under LDS budgets $1 (KB list)
for kb in $1; do
  echo -n "LDS_KB=$kb:"
  for l in 2 5 8 10 13 15 18 22 24 27; do
    QNNP_GFX950_DW_LDS_KB=$kb python bench.py --layer $l --dw-kernel 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' L%d %.1f' % (d['layer'], d['ms']*1e3), end='')"
  done
  echo
done
