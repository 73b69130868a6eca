#!/bin/bash
# lane forms of the requantization (staged / long-K pointwise kernels, weight-stationary 3x3 convolution): the GPU tier,
# then a same-box A/B of the library before / after through bench.py
TAG=${1:-r03lane}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider 2>&1 | tail -n 40 > $OUT/pytest_gpu.log; tail -n 12 $OUT/pytest_gpu.log
cp qnnpack_amd/libqnnpack_gfx950.so qnnpack_amd/libqnnpack_gfx950_lane.so
bash scripts/gpu_ab_lib.sh $TAG qnnpack_amd/libqnnpack_gfx950_prelane.so qnnpack_amd/libqnnpack_gfx950_lane.so 2>&1 | tee $OUT/ab.txt
