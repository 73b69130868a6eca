#!/bin/bash
export QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so
for shape in "55 55 3 1 1 16 64" "55 55 3 1 1 32 128" "27 27 3 1 1 32 128" "27 27 3 1 1 48 192" "13 13 3 1 1 48 192" "27 27 3 1 1 64 256"; do
  for pc in 1 2 3; do echo "per_cu $pc"; QNNP_WS16S_PER_CU=$pc timeout 100 python tools/conv_one_time.py $shape 3 32 2>&1 | tail -1; done
done
