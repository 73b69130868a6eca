#!/bin/bash
# measurement only: same-box A/B of two builds of the library (qnnpack_amd/libqnnpack_gfx950.so.A / .B) on one
# bench.py command ($1 = bench arguments, e.g. "--layer 99"); the in-tree .so is restored to B at the end.
export TMPDIR=/tmp; mkdir -p gpurun_out/ab
L=qnnpack_amd/libqnnpack_gfx950.so
for rep in 1 2 3; do
  for v in A B; do
    cp $L.$v $L
    echo -n "$v "; timeout 300 python bench.py $1 --steps 50 --warmup 10 --no-cpu-baseline 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('kernel'), d.get('ms', d.get('ms_per_step')), d.get('tops', d.get('value')))"
  done
done | tee gpurun_out/ab/libab.txt
cp $L.B $L
