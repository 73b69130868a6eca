#!/bin/bash
# round 5: 7x7 entry layer, 8-byte loads for the second half of a 32-byte slot -- the candidate library (built beside the product one
# as libqnnpack_gfx950_abl.so) against the product library on the same box: parity first, then the layer at batch 128, interleaved
TAG=${1:-r05c3n}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
NEW=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so
QNNP_GFX950_LIBRARY=$NEW timeout 900 python -m pytest tests/test_gpu_convc3rows.py -q -p no:cacheprovider 2>&1 | tail -n 6 | tee $OUT/pytest.log
QNNP_GFX950_LIBRARY=$NEW timeout 900 python -m pytest tests/test_gpu_conv_lists_bench_batch.py -q -p no:cacheprovider -k "k7" 2>&1 | tail -n 4 | tee -a $OUT/pytest.log
for rep in 1 2; do
  echo "product" | tee -a $OUT/conv7x7.txt
  timeout 200 python tools/conv_one_time.py 224 224 7 2 1 3 64 3 2>&1 | grep -v amdgpu.ids | tee -a $OUT/conv7x7.txt
  echo "candidate" | tee -a $OUT/conv7x7.txt
  QNNP_GFX950_LIBRARY=$NEW timeout 200 python tools/conv_one_time.py 224 224 7 2 1 3 64 3 2>&1 | grep -v amdgpu.ids | tee -a $OUT/conv7x7.txt
done
