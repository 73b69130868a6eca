#!/bin/bash
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -n 6
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 > $OUT/bench.json
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read()); e=d["extra"]; s=e["mobilenetv2_sweep"]
print(d["value"], d["roofline"]["frac"], d["roofline"].get("vendor_int8_gemm_no_epilogue_tops"), "conv", e["q8conv_3x3_56x56x64_b128"]["ms"], "dw", e["q8dwconv_mobilenetv2_layers"]["ms"], "sweep", s["images_per_s"], "net", e["mobilenetv2_network"]["images_per_s"])
print(" ".join(f"{r['layer']}:{r['ms']*1000:.1f}" for r in s["layers"]))
PY
