#!/bin/bash
# round 5: the packed shift >= 1 requantization tail (kRqBoundedLanePk) -- parity on the kernels that take it, then the
# MobileNetV2 sweep at requantization scale 0.0125 and 0.5: product library against the previous build (OLD) on the same box
TAG=${1:-r05rqpk}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
OLD=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so
timeout 1500 python -m pytest tests/test_gpu_requant_packed_tail.py tests/test_gpu_sweep_bench_batch.py tests/test_gpu_convstream.py tests/test_gpu_convc3rows.py tests/test_gpu_requant_corners.py -q -p no:cacheprovider -x 2>&1 | tail -n 8 | tee $OUT/pytest.log
for rep in 1 2; do
  echo "== product (packed tail)" | tee -a $OUT/sweep.txt
  timeout 300 python tools/realistic_scale_time.py 20 0.5 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweep.txt
  echo "== previous build" | tee -a $OUT/sweep.txt
  QNNP_GFX950_LIBRARY=$OLD timeout 300 python tools/realistic_scale_time.py 20 0.5 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweep.txt
done
