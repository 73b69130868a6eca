#!/bin/bash
# GPU tier + the sweep / lists part of the bench (no profiles)
TAG=${1:-r06e}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider 2>&1 | tail -n 60 > $OUT/pytest_gpu.log
tail -n 30 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 --full-out $OUT/bench_full.json 2> $OUT/bench.err | tail -n 1 > $OUT/bench.json
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print(json.dumps({k: d[k] for k in ("value", "ms_per_step")}), d["roofline"]["frac"], json.dumps(d["roofline"]["secondary"]))
f = json.load(open("$OUT/bench_full.json"))
for net, v in f["extra"].get("conv_lists", {}).items():
    print(net, v["images_per_s_by_sum_of_layers"], v["frac_of_bound"])
print([ (r["layer"], r["kernel"].replace("q8_",""), round(r["ms"]*1e3,1)) for r in f["extra"]["mobilenetv2_sweep"]["layers"]])
PY
