#!/bin/bash
# scratch experiments of the moment (same box): gpurun -- bash scripts/gpu_exp.sh <tag>
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cp qnnpack_amd/libqnnpack_gfx950.so /tmp/keep.so
layer() { timeout 120 python bench.py --layer $1 --steps 20 --warmup 3 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['layer'], d['kernel'], round(d['ms']*1000,1), 'us', d['gbs'])" | tee -a $OUT/exp.txt; }
cp tmp_libs/abl.so qnnpack_amd/libqnnpack_gfx950.so
for Y in 3 4 7 12 17 21 26 11 14 16; do
  layer $Y auto
  for B in 2 3 4 5 6 7 8; do QNNP_PW_BLOCKS=$B layer $Y blocks$B; done
done
cp /tmp/keep.so qnnpack_amd/libqnnpack_gfx950.so
