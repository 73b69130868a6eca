#!/bin/bash
# residual add folded into the project convolutions: parity, then the network with / without the fold
TAG=${1:-resid}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_residual.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 12 | tee $OUT/pytest.log
echo "== network, per operator"
timeout 200 python tools/network_profile.py 128 > $OUT/network_plain.txt 2>&1; echo "rc=$?"; head -n 3 $OUT/network_plain.txt
timeout 200 python tools/network_profile.py 128 fold > $OUT/network_fold.txt 2>&1; echo "rc=$?"; head -n 3 $OUT/network_fold.txt
grep -E "project|add" $OUT/network_plain.txt | head -n 40
echo "-- folded"
grep -E "project" $OUT/network_fold.txt | head -n 40
