#!/bin/bash
export TMPDIR=/tmp
run() { python bench.py --layer $1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' L%d %.1f' % (d['layer'], d['ms']*1e3), end='')"; }
cp qnnpack_amd/libqnnpack_gfx950.so /tmp/keep.so
for rep in 1 2 3; do
for lib in h_final i_prio; do cp tmp_libs/$lib.so qnnpack_amd/libqnnpack_gfx950.so
  echo -n "$lib:"; for l in 2 8 13 18 22; do run $l; done; echo
done; done
cp /tmp/keep.so qnnpack_amd/libqnnpack_gfx950.so
