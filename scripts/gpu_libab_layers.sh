#!/bin/bash
# measurement only: same-box A/B of two library builds (.so.A / .so.B) over a list of sweep layers ($1), --dw-kernel $2
export TMPDIR=/tmp; mkdir -p gpurun_out/ab
L=qnnpack_amd/libqnnpack_gfx950.so
for layer in $1; do
  for v in A B A B; do
    cp $L.$v $L
    echo -n "layer $layer $v "; timeout 120 python bench.py --layer $layer --dw-kernel ${2:-0} --steps 40 --warmup 10 --no-cpu-baseline 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel'], d['ms'], d['gbs'])"
  done
done | tee gpurun_out/ab/libab_layers.txt
cp $L.A $L
