#!/bin/bash
# dilated column walk: parity (new file + the kernel-G regression files), then timing against the LDS-tiled kernel
TAG=${1:-r04dil}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dwcol_dilated.py tests/test_gpu_dwcol.py tests/test_gpu_dwconv_matrix.py tests/test_gpu_operators.py -q -p no:cacheprovider -x 2>&1 | tail -n 15 | tee $OUT/pytest.log
timeout 600 python tools/dw_dil_time.py 128 0 2 2>&1 | grep -v amdgpu.ids | tee $OUT/dw_dil_time.txt
