#!/bin/bash
# measurement only: A/B the depthwise kernels ($1, default "2 3 4") per sweep layer inside ONE call
# (box-to-box variance is +-10 %)
for rep in 1 2; do
for k in ${1:-2 3 4}; do
  echo -n "dw_kernel=$k:"
  for l in 2 5 8 10 13 15 18 22 24 27; do
    python bench.py --layer $l --dw-kernel $k 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' L%d %.1f' % (d['layer'], d['ms']*1e3), end='')"
  done
  echo
done
done
