#!/bin/bash
# streaming (nt) cache hints on activation loads / output stores (hip/stream_policy.h): GPU tier on the new library, then
# a same-box A/B against the library without the hints through bench.py
TAG=${1:-r03nt}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider 2>&1 | tail -n 40 > $OUT/pytest_gpu.log; tail -n 4 $OUT/pytest_gpu.log
cp qnnpack_amd/libqnnpack_gfx950.so qnnpack_amd/libqnnpack_gfx950_nt.so
bash scripts/gpu_ab_lib.sh $TAG qnnpack_amd/libqnnpack_gfx950_plain.so qnnpack_amd/libqnnpack_gfx950_nt.so 2>&1 | tee $OUT/ab.txt
