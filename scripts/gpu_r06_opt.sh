#!/bin/bash
TAG=${1:-r06k}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 build_tools/ubench_mfma2 2>&1 | grep -A40 "random" | tee $OUT/ubench_mfma2_orders.txt
export QNNP_GFX950_LIBRARY=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so
timeout 600 python tools/gemm_ab.py --variants 23 --c16-opts 0,1 --rounds 9 2>&1 | grep -v amdgpu.ids | tee $OUT/gemm_ab_c16_snake.txt
