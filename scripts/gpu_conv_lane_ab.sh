#!/bin/bash
# 3x3 convolution (configs[2]) and two scales: the library before the lane forms against the current one, alternating
TAG=${1:-r03convlane}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cp qnnpack_amd/libqnnpack_gfx950.so /tmp/new.so
for r in 1 2 3 4; do
  for L in prelane new; do
    if [ $L = prelane ]; then cp qnnpack_amd/libqnnpack_gfx950_prelane.so qnnpack_amd/libqnnpack_gfx950.so; else cp /tmp/new.so qnnpack_amd/libqnnpack_gfx950.so; fi
    a=$(timeout 120 python bench.py --layer 99 --steps 40 --warmup 5 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel'], d['ms'])")
    b=$(timeout 120 python bench.py --layer 99 --steps 40 --warmup 5 --out-scale 20 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms'])")
    echo "round $r $L: bench scale $a   realistic scale (--out-scale 20) $b" | tee -a $OUT/conv_ab.txt
  done
done
cp /tmp/new.so qnnpack_amd/libqnnpack_gfx950.so
