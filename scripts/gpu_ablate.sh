#!/bin/bash
# measurement only: time the 4096^3 GEMM under each ablation mask (library built with ABLATION=1)
export TMPDIR=/tmp; mkdir -p gpurun_out/$1
if [ -n "$2" ]; then timeout 900 python -m pytest $2 -q --maxfail=20 -p no:cacheprovider 2>&1 | tail -n 30; fi
for a in ${3:-0 1 2 4 8 12 15 27 31}; do
  echo -n "ablate=$a "; QNNP_GFX950_ABLATE=$a timeout 300 python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['launch_ms'], d['ms_per_step'], d['roofline']['achieved'])"
done | tee gpurun_out/$1/ablate.txt
