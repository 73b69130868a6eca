#!/bin/bash
# round 6: the LDS-staged 7x7 / 5x5 entry-layer kernel -- parity, then A/B against the register-path kernel on the reference's shapes
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_convc3rows.py tests/test_gpu_requant_packed_tail.py -x -q 2>&1 | tail -5 > gpurun_out/c3lds_pytest.log
{
for v in 14 30 14 30; do
  timeout 120 python tools/conv_one_time.py 224 224 7 2 1 3 64 5 $v | tail -3
done
for v in 14 30; do
  timeout 120 python tools/conv_one_time.py 224 224 7 2 1 3 96 5 $v | tail -2
done
KZP=126 timeout 120 python tools/conv_one_time.py 224 224 7 2 1 3 64 5 30 | tail -2
KZP=126 timeout 120 python tools/conv_one_time.py 224 224 7 2 1 3 64 5 14 | tail -2
QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so timeout 200 python tools/trace_c3lds.py
} 2>&1 | grep -v amdgpu.ids > gpurun_out/c3lds_ab.txt
cat gpurun_out/c3lds_pytest.log gpurun_out/c3lds_ab.txt
