#!/bin/bash
# measurement only (ABLATION=1 build): what each part of the LDS-tiled convolution costs on bench.py --layer 99
export TMPDIR=/tmp; mkdir -p gpurun_out/ab
for a in 0 1 2 4 7 8 16 32 48 55 63 0; do
  export QNNP_CONV_ABL=$a
  echo -n "abl=$a "; timeout 300 python bench.py --layer 99 --steps 50 --warmup 10 --no-cpu-baseline 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel'], d['ms'], d['tops'])"
done | tee gpurun_out/ab/convabl.txt
