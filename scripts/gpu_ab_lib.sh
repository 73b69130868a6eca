#!/bin/bash
# same-box A/B of two builds of the library: gpurun -- bash scripts/gpu_ab_lib.sh <tag> <a.so> <b.so>   (paths in the repo)
TAG=${1:-ab}; A=$2; B=$3; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cp qnnpack_amd/libqnnpack_gfx950.so /tmp/keep.so
summ() {
python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read())
e=d["extra"]; s=e["mobilenetv2_sweep"]
print({k:d[k] for k in ("value",)}, d["roofline"]["sustained_launch_ms"], "conv3x3", e["q8conv_3x3_56x56x64_b128"]["ms"], "dw", e["q8dwconv_mobilenetv2_layers"]["ms"],
      "sweep", s["images_per_s"], "net", e["mobilenetv2_network"]["images_per_s"], "folded", e.get("mobilenetv2_network_adds_folded", {}).get("images_per_s"),
      "deconv3x3s2", e.get("next_rows", {}).get("q8deconv_3x3s2_28x28x64_32", {}).get("ms"))
print(" ".join(f"{r['layer']}:{r['ms']*1000:.1f}" for r in s["layers"]))
PY
}
for round in 1 2; do
  for L in $A $B; do
    cp $L qnnpack_amd/libqnnpack_gfx950.so
    timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -n 1 > $OUT/bench_$(basename $L .so)_$round.json
    echo "== $L round $round"; summ $OUT/bench_$(basename $L .so)_$round.json
  done
done
cp /tmp/keep.so qnnpack_amd/libqnnpack_gfx950.so
