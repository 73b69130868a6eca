#!/bin/bash
for shape in "14 14 1 1 1 64 384" "14 14 1 1 1 96 576" "7 7 1 1 1 160 960" "7 7 1 1 1 576 160" "7 7 1 1 1 960 160" "14 14 1 1 1 192 64" "28 28 1 1 1 32 192" "28 28 1 1 1 144 32"; do
  for v in 0 29 24; do timeout 100 python tools/conv_one_time.py $shape 3 $v 2>&1 | tail -1; done
done
