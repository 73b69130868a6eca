#!/bin/bash
# stride-2 deconvolution streaming kernel: parity, then timing against the phase-table GEMMs
TAG=${1:-deconv}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_deconvolution.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 15 | tee $OUT/pytest.log
timeout 300 python tools/next_rows_time.py 128 1 0 2>&1 | grep -v amdgpu.ids | tee $OUT/deconv_ab.txt
