#!/bin/bash
# scratch experiments of the moment (same box): gpurun -- bash scripts/gpu_exp.sh <tag>
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cp qnnpack_amd/libqnnpack_gfx950.so /tmp/keep.so
layer() { timeout 120 python bench.py --layer $1 --steps 20 --warmup 3 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['layer'], d['kernel'], round(d['ms']*1000,1), 'us', d['gbs'])" | tee -a $OUT/exp.txt; }
timeout 600 python -m pytest tests/test_gpu_pwstream.py tests/test_gpu_sweep_bench_batch.py -q -p no:cacheprovider 2>&1 | tail -n 3
for L in new7 new8; do cp tmp_libs/$L.so qnnpack_amd/libqnnpack_gfx950.so; for Y in 19 20 23 25 28 29; do layer $Y $L; done; done
cp /tmp/keep.so qnnpack_amd/libqnnpack_gfx950.so
