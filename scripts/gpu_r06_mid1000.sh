#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_gemm128x.py -q 2>&1 | tail -3
for v in 0 24 25 26 29 15 23; do timeout 100 python tools/conv_one_time.py 13 13 1 1 1 512 1000 3 $v 2>&1 | tail -1; done
for v in 0 24 29; do timeout 100 python tools/conv_one_time.py 14 14 1 1 1 512 1000 3 $v 2>&1 | tail -1; done
