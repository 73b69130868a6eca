#!/bin/bash
TAG=${1:-r06n}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dwmfma16.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 25 | tee $OUT/pytest_dwm16.log
timeout 600 python tools/dw_m16_time.py 3 2>&1 | grep -v amdgpu.ids | tee $OUT/dw_m16_time.txt
