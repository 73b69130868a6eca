#!/bin/bash
# measurement only (ABLATION=1 build): matrix-core LDS depthwise kernel with stores / loads removed
for a in 0 1 2 3; do
  echo -n "QNNP_DW_ABL=$a:"
  for l in 2 5 8 13; do
    QNNP_DW_ABL=$a python bench.py --layer $l --dw-kernel 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' L%d %.1f' % (d['layer'], d['ms']*1e3), end='')"
  done
  echo
done
