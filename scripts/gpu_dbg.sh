#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out/r03dbg
timeout 120 python -m pytest tests/test_gpu_conv_matrix.py -x -s -q -p no:cacheprovider -k "test_k_gt_step_subtile and 33-m1-17" -p no:faulthandler > gpurun_out/r03dbg/s.log 2>&1
grep -v "^  File" gpurun_out/r03dbg/s.log | cut -c1-400 | tail -n 25
