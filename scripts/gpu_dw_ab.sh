#!/bin/bash
# depthwise kernels: parity of the column walks, then alternating timings of "dwconv_kernel" variants over sweep layers
#   gpu_dw_ab.sh <tag> "<variants>" "<layers>"
TAG=${1:-dwab}; VARS=${2:-"0 6"}; LAYERS=${3:-"2 8 13 18 22 27"}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_dwcol.py tests/test_gpu_dwconv_matrix.py tests/test_gpu_fullsize.py -m gpu -q -x -k "not q8gemm and not c5 and not c3" -p no:cacheprovider 2>&1 | tail -n 8 | tee $OUT/pytest.log
echo "== timing (alternating)"
for round in 1 2; do
  for L in $LAYERS; do
    for V in $VARS; do
      timeout 120 python bench.py --layer $L --steps 30 --warmup 5 --dw-kernel $V 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('layer',d['layer'],'variant $V',d['kernel'],round(d['ms']*1000,2),'us',d['gbs'],'GB/s')" | tee -a $OUT/dw_timing.txt
    done
  done
done
