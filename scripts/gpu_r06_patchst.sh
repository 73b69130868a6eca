#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_convpatch.py tests/test_gpu_convpatch_chunked.py tests/test_gpu_conv_lists_bench_batch.py -q 2>&1 | tail -4
for shape in "28 28 3 1 1 128 128" "14 14 3 1 1 256 256" "7 7 3 1 1 512 512" "56 56 3 2 1 128 128" "14 14 3 2 1 512 512" "112 112 3 1 1 128 128" "56 56 3 1 1 256 256" "112 112 3 1 1 64 128"; do
  timeout 100 python tools/conv_one_time.py $shape 4 0 2>&1 | tail -2
done
