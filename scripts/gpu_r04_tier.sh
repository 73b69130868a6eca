#!/bin/bash
# the whole GPU tier (+ smoke), log under gpurun_out/<tag>/
TAG=${1:-r04tier}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -n 40 | tee $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 3 | tee $OUT/smoke.log
