#!/bin/bash
# Round 6, first call: MFMA feeding microbenchmark, the rows_per_image == 1 cases, the bench line (contract line without `extra`).
TAG=${1:-r06a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== device"; rocminfo 2>/dev/null | grep -E "gfx|Compute Unit" | head -4; nproc
echo "== ubench_mfma2"
timeout 300 build_tools/ubench_mfma2 2>&1 | tee $OUT/ubench_mfma2.txt
echo "== pytest (changed files)"
timeout 900 python -m pytest tests/test_gpu_pwstream.py tests/test_gpu_gemm256c.py -m gpu -q -p no:cacheprovider 2>&1 | tail -n 8 | tee $OUT/pytest_changed.log
echo "== bench"
timeout 900 python bench.py --steps 20 --warmup 5 --full-out $OUT/bench_full.json 2> $OUT/bench.err > $OUT/bench.stdout
tail -n 1 $OUT/bench.stdout > $OUT/bench.json
wc -c $OUT/bench.json
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print(json.dumps({k: d[k] for k in ("value", "ms_per_step")}), json.dumps(d["roofline"], indent=0).replace("\n", " "))
print(d["cpu_baseline"])
PY
tail -n 5 $OUT/bench.err
