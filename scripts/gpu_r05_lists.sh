#!/bin/bash
# round 5: the ResNet / ShuffleNet shape lists -- parity at the bench batch against the compiled reference, then per-layer times
TAG=${1:-r05lists}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_conv_lists_bench_batch.py -q -p no:cacheprovider 2>&1 | tail -n 15 | tee $OUT/pytest.log
timeout 600 python tools/conv_lists_time.py ${2:-all} 2>&1 | grep -v amdgpu.ids | tee $OUT/conv_lists.txt
