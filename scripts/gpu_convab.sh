#!/bin/bash
# measurement only: A/B of the LDS-tiled convolution flavours (bench.py --layer 99)
export TMPDIR=/tmp; mkdir -p gpurun_out/ab
if [ -n "$1" ]; then timeout 600 python -m pytest $1 -q -m gpu --maxfail=10 -p no:cacheprovider 2>&1 | tail -n 3; fi
for rep in 1 2 3; do
  for e in 0 1; do
    if [ $e = 1 ]; then export QNNP_CONV_NOOSTG=1; else unset QNNP_CONV_NOOSTG; fi
    echo -n "noostg=$e "; timeout 300 python bench.py --layer 99 --steps 50 --warmup 10 --no-cpu-baseline 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel'], d['ms'], d['tops'])"
  done
done | tee gpurun_out/ab/convab.txt
