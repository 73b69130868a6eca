#!/bin/bash
# fused strip kernel: parity, per-operator times, plans and cycle stamps (ablation build)
TAG=${1:-r04j}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_fused.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 8 | tee $OUT/pytest_fused.log
timeout 300 python tools/network_profile.py 128 fuse 2>&1 | grep -v amdgpu.ids | tee $OUT/network_per_operator_fused.txt | head -24
QNNP_GFX950_PRINT_PLAN=1 QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so timeout 300 python tools/trace_fused.py ${2:-b0_fused b1_fused b2_fused b4_fused b7_fused b14_fused} 2>&1 | grep -v amdgpu.ids | sort -u | tee $OUT/trace_fused.txt
