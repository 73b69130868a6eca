#!/bin/bash
# measurement only: optional tests ($1), then the MobileNetV2 sweep per-layer table
export TMPDIR=/tmp
if [ -n "$1" ]; then timeout 900 python -m pytest $1 -q -x -p no:cacheprovider 2>&1 | tail -n 8; fi
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); e=d['extra']['mobilenetv2_sweep']
print('gemm', d['value'], 'sweep img/s', e['images_per_s'], 'ms', e['ms_per_batch'], 'c3', d['extra']['q8conv_3x3_56x56x64_b128']['ms'], 'dw', d['extra']['q8dwconv_mobilenetv2_layers'])
for l in e['layers']: print(l['layer'], l['shape'], l['kernel'], l['ms'], l['gbs'])
"
