#!/bin/bash
# round 5: the 32-byte-slot first-layer kernel -- parity, then ResNet's 7x7 entry layer at batch 128
TAG=${1:-r05c3}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_convc3rows.py tests/test_gpu_convstream.py -q -p no:cacheprovider 2>&1 | tail -n 12 | tee $OUT/pytest.log
timeout 900 python -m pytest tests/test_gpu_conv_lists_bench_batch.py -q -p no:cacheprovider -k "k7" 2>&1 | tail -n 4 | tee -a $OUT/pytest.log
timeout 200 python tools/conv_one_time.py 224 224 7 2 1 3 64 3 2>&1 | grep -v amdgpu.ids | tee $OUT/conv7x7.txt
timeout 200 python tools/conv_one_time.py 224 224 3 2 1 3 32 3 2>&1 | grep -v amdgpu.ids | tee -a $OUT/conv7x7.txt
