#!/bin/bash
# round 5: one iteration on the patch kernel -- parity of the dense ResNet rows, their times, stamps
TAG=${1:-r05piter}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_lists_bench_batch.py -q -p no:cacheprovider -x -k "k3s" 2>&1 | tail -n 4 | tee $OUT/pytest.log
timeout 600 python tools/conv_lists_time.py dense3x3 2>&1 | grep -v amdgpu.ids | tee $OUT/conv_lists.txt
QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so timeout 200 python tools/trace_patch.py 14 14 3 1 1 256 256 2>&1 | grep -v amdgpu.ids | tee -a $OUT/patch_stamps.txt
