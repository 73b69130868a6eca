#!/bin/bash
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cp qnnpack_amd/libqnnpack_gfx950.so /tmp/keep.so
timeout 600 python -m pytest tests/test_gpu_convwave.py tests/test_gpu_fullsize.py -q -x -p no:cacheprovider 2>&1 | tail -n 3
for L in new15 new16 new15 new16; do cp tmp_libs/$L.so qnnpack_amd/libqnnpack_gfx950.so
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['extra']['q8conv_3x3_56x56x64_b128'])"
done
cp tmp_libs/abl.so qnnpack_amd/libqnnpack_gfx950.so
timeout 300 python tools/trace_dump.py 99 2>&1 | tail -n 6 | tee $OUT/trace_conv.txt
cp /tmp/keep.so qnnpack_amd/libqnnpack_gfx950.so
