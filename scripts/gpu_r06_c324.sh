#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_convc3rows.py tests/test_gpu_reference_lists.py -q -k "c3rows or 224x224 or 24 or row_slot or wide" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_convc3rows.py -q 2>&1 | tail -3
for i in 1 2; do timeout 100 python tools/conv_one_time.py 224 224 3 2 1 3 24 4 0 2>&1 | tail -2; done
