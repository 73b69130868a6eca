for shape in "56 56 1 1 1 24 88" "28 28 1 1 1 88 88" "56 56 1 1 1 24 36" "28 28 1 1 1 144 36" "28 28 1 1 1 144 72" "28 28 1 1 1 24 88" "56 56 1 1 1 24 24" "55 55 1 1 1 96 16"; do
  for v in 5 29; do timeout 100 python tools/conv_one_time.py $shape 3 $v 2>&1 | tail -1; done
done
