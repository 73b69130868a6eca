#!/bin/bash
for shape in "14 14 1 1 1 256 1024" "28 28 1 1 1 128 512" "28 28 1 1 1 512 256" "14 14 1 1 1 1024 512" "56 56 1 1 1 256 128" "56 56 1 1 1 64 256"; do
  for v in 0 24 25 26 23 5; do timeout 100 python tools/conv_one_time.py $shape 3 $v 2>&1 | tail -1; done
done
