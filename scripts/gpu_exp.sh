#!/bin/bash
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp


timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -n 15
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 > $OUT/bench.json
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read()); e=d["extra"]; s=e["mobilenetv2_sweep"]
print(d["value"], "sweep", s["images_per_s"], "net", e["mobilenetv2_network"]["images_per_s"])
print(" ".join(f"{r['layer']}:{r['ms']*1000:.1f}" for r in s["layers"]))
PY
