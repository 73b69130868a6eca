#!/bin/bash
# strip kernel with the chunk's weights staged in LDS: parity, then per-operator times with / without ("fused_weights" 2)
TAG=${1:-r04w}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_gemm256c.py -m gpu -q -p no:cacheprovider 2>&1 | tail -n 8 | tee $OUT/pytest_fused.log
timeout 300 python tools/network_profile.py 128 fuse 2>&1 | grep -v amdgpu.ids | tee $OUT/network_per_operator_fused.txt | head -24
QNNP_FUSED_WEIGHTS=2 timeout 300 python tools/network_profile.py 128 fuse 2>&1 | grep -v amdgpu.ids | tee $OUT/network_per_operator_fused_weights_from_l2.txt | head -24
timeout 300 python tools/network_profile.py 128 fuse 2>&1 | grep -v amdgpu.ids | tee $OUT/network_per_operator_fused_2.txt | head -3
