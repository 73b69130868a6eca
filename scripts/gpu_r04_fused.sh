#!/bin/bash
# round 4: the fused-block strip kernel: parity (network vs oracle, all blocks fused, forced strip heights, bench batch vs the
# stand-alone chain), then per-operator times of the fused network beside the plain / adds-folded ones
TAG=${1:-r04f}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== parity"
timeout 1500 python -m pytest tests/test_gpu_fused.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 25 | tee $OUT/pytest_fused.log
echo "== per operator, fused"
timeout 300 python tools/network_profile.py 128 fuse 2>&1 | grep -v amdgpu.ids | tee $OUT/network_per_operator_fused.txt | head -40
echo "== per operator, adds folded"
timeout 300 python tools/network_profile.py 128 fold 2>&1 | grep -v amdgpu.ids | tee $OUT/network_per_operator_folded.txt | head -3
