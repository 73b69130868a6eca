#!/bin/bash
# round 6: the small-channel weight-stationary 3x3 kernel: parity, then SqueezeNet's fire-module rows against what they ran on
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_convws16s.py -x -q 2>&1 | tail -15 > gpurun_out/ws16s_pytest.log
cat gpurun_out/ws16s_pytest.log
{
for shape in "55 55 3 1 1 16 64" "55 55 3 1 1 32 128" "27 27 3 1 1 32 128" "27 27 3 1 1 48 192" "13 13 3 1 1 48 192" "27 27 3 1 1 64 256" "13 13 3 1 1 64 256" "56 56 3 1 1 64 64"; do
  for v in 1 3 22 8 32 0; do timeout 100 python tools/conv_one_time.py $shape 3 $v 2>&1 | tail -1; done
done
} 2>&1 | grep -v "amdgpu.ids\|Traceback\|File \|raise\|layer = \|lib.run" > gpurun_out/ws16s_time.txt
cat gpurun_out/ws16s_time.txt
