#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out/$1
timeout 900 python -m pytest tests/test_gpu_convlds.py tests/test_gpu_fullsize.py::test_c3_q8conv_3x3_56x56x64_batch128 -q --maxfail=10 -p no:cacheprovider 2>&1 | tail -n 6
timeout 300 python tools/trace_dump.py 99 2>&1 | tail -n 7
timeout 300 python bench.py --layer 99 --steps 30 --warmup 5 2>/dev/null | tail -n 1
