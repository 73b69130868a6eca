#!/bin/bash
# round 5: the channel-chunked patch flavour (stride-2 3x3 with 256 / 512 channels) -- parity of the dense 3x3 rows at the
# bench batch, then the measurement build with the chunked plan off / automatic / forced on the same box
TAG=${1:-r05chunk}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ABL=$PWD/qnnpack_amd/libqnnpack_gfx950_abl.so
timeout 1500 python -m pytest tests/test_gpu_convpatch_chunked.py tests/test_gpu_conv_lists_bench_batch.py -q -p no:cacheprovider -k "k3 or chunk or tap" 2>&1 | tail -n 8 | tee $OUT/pytest.log
for mode in 0 -1 1 0 -1; do
  echo "== QNNP_PATCH_CHUNK=$mode" | tee -a $OUT/dense3x3.txt
  QNNP_GFX950_LIBRARY=$ABL QNNP_PATCH_CHUNK=$mode timeout 300 python tools/conv_lists_time.py dense3x3 2>&1 | grep -v amdgpu.ids | tee -a $OUT/dense3x3.txt
done
