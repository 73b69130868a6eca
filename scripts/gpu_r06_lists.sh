#!/bin/bash
TAG=${1:-r06f}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest reference lists + the name-expectation updates"
timeout 2400 python -m pytest tests/test_gpu_reference_lists.py tests/test_gpu_sweep_bench_batch.py tests/test_gpu_pwstream.py tests/test_gpu_requant_packed_tail.py tests/test_gpu_gemm128x.py -m gpu -q --maxfail=30 --durations=8 -p no:cacheprovider 2>&1 | tail -n 60 | tee $OUT/pytest_lists.log
echo "== bench (all lists)"
timeout 1200 python bench.py --steps 20 --warmup 5 --full-out $OUT/bench_full.json 2> $OUT/bench.err | tail -n 1 > $OUT/bench.json
wc -c $OUT/bench.json
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print(json.dumps({k: d[k] for k in ("value", "ms_per_step")}), d["roofline"]["frac"])
f = json.load(open("$OUT/bench_full.json"))
for net, v in f["extra"].get("reference_bench_lists", {}).items():
    print(net, v["images_per_s_by_sum_of_layers"], v["frac_of_bound"], "worst", v["worst_row"])
    for r in v["layers"]:
        if r[-1] < 0.12: print("    ", r)
PY
tail -n 3 $OUT/bench.err
