#!/bin/bash
# round 6: LDS-staged entry-layer kernel, pairs-per-band sweep (measurement build) against the register-path kernel
mkdir -p gpurun_out
export QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so
{
for rep in 1 2; do
  echo "== register path"; timeout 120 python tools/conv_one_time.py 224 224 7 2 1 3 64 6 14 | tail -3
  for ppb in 1 2 4 7 8; do
    echo "== lds ppb $ppb"; QNNP_C3L_PPB=$ppb timeout 120 python tools/conv_one_time.py 224 224 7 2 1 3 64 6 30 | tail -3
  done
done
echo "== 96 channels"; timeout 120 python tools/conv_one_time.py 224 224 7 2 1 3 96 6 14 | tail -2
for ppb in 2 4; do QNNP_C3L_PPB=$ppb timeout 120 python tools/conv_one_time.py 224 224 7 2 1 3 96 6 30 | tail -2; done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/c3lds_ppb.txt
cat gpurun_out/c3lds_ppb.txt
