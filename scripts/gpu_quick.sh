#!/bin/bash
# quick GPU iteration: selected tests + bench variants.
# usage: gpurun -- bash scripts/gpu_quick.sh <tag> "<pytest args>" "<bench args 1>;<bench args 2>;..."
TAG=${1:-q}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ -n "$2" ]; then timeout 900 python -m pytest $2 -q --maxfail=20 -p no:cacheprovider 2>&1 | tail -n 60 | tee $OUT/pytest.log; fi
if [ -n "$3" ]; then
  IFS=';' read -ra VARS <<< "$3"
  i=0
  for v in "${VARS[@]}"; do
    echo "== bench $v"
    timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $v 2>&1 | tail -n 1 | tee $OUT/bench_$i.json | cut -c1-1000
    i=$((i+1))
  done
fi
