#!/bin/bash
# measurement only: depthwise layers of the sweep on the default kernel vs the matrix-core LDS kernel (dwconv_kernel=5)
export TMPDIR=/tmp; mkdir -p gpurun_out/ab
for layer in 8 10 13 15 18 22; do
  for k in 0 5; do
    echo -n "layer $layer dw_kernel=$k "; timeout 120 python bench.py --layer $layer --dw-kernel $k --steps 40 --warmup 10 --no-cpu-baseline 2>&1 | tail -n 1 | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['kernel'], d['ms'], d['gbs'])
except Exception as e: print('failed', e)"
  done
done | tee gpurun_out/ab/dwf.txt
