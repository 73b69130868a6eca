#!/bin/bash
# measurement only: A/B the GEMM kernel flavours on 4096^3 (gemm_kernel option values in $2), after running tests $1
export TMPDIR=/tmp; mkdir -p gpurun_out/ab
if [ -n "$1" ]; then timeout 900 python -m pytest $1 -q --maxfail=20 -p no:cacheprovider 2>&1 | tail -n 30; fi
for rep in 1 2; do
for v in ${2:-2 4}; do
  echo -n "gemm_kernel=$v "; timeout 300 python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline --gemm-kernel $v 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['launch_ms'], d['ms_per_step'], d['roofline']['achieved'])"
done
done | tee gpurun_out/ab/ab.txt
