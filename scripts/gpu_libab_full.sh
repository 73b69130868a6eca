#!/bin/bash
# measurement only: same-box A/B of two library builds (.so.A / .so.B) on the whole bench line (GEMM + sweep + conv)
export TMPDIR=/tmp; mkdir -p gpurun_out/ab
L=qnnpack_amd/libqnnpack_gfx950.so
for rep in 1 2 3; do
  for v in A B; do
    cp $L.$v $L
    echo -n "$v "; timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -n 1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); e=d['extra']; s=e['mobilenetv2_sweep']
print('gemm', d['value'], 'launch_ms', d['roofline']['launch_ms'], '| sweep img/s', s['images_per_s'], 'sum_ms', s['sum_of_layer_ms'], '| conv ms', e['q8conv_3x3_56x56x64_b128']['ms'], '| dw ms', e['q8dwconv_mobilenetv2_layers']['ms'])"
  done
done | tee gpurun_out/ab/libab_full.txt
cp $L.B $L
