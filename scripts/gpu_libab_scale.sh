#!/bin/bash
# measurement only: same-box A/B of two library builds over sweep layers ($1) at output scale $2 (requant = 0.25 / scale)
export TMPDIR=/tmp; mkdir -p gpurun_out/ab
L=qnnpack_amd/libqnnpack_gfx950.so
for layer in $1; do
  for v in A B A B; do
    cp $L.$v $L
    echo -n "layer $layer $v "; timeout 120 python bench.py --layer $layer --out-scale $2 --steps 40 --warmup 10 --no-cpu-baseline 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel'], d['ms'], d['gbs'])"
  done
done | tee gpurun_out/ab/libab_scale.txt
cp $L.B $L
