#!/bin/bash
# usage: gpurun -- bash scripts/gpu_quick2.sh <tag> [regen]   -- GPU tier (+ sweep dispatch table refresh), bench line
TAG=${1:-q}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ "$2" = regen ]; then
  QNNP_WRITE_SWEEP_KERNELS=1 timeout 600 python -m pytest tests/test_gpu_sweep_bench_batch.py -q -p no:cacheprovider 2>&1 | tail -n 3
  cp tests/golden/sweep_kernels.json $OUT/
fi
timeout 1200 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider 2>&1 | tail -n 60 > $OUT/pytest_gpu.log
tail -n 12 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 ${BENCH_ARGS} 2>$OUT/bench.err | tail -n 1 > $OUT/bench.json
tail -n 3 $OUT/bench.err
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read())
print({k:d[k] for k in ("value","ms_per_step","n_gpus")}, {k:d["roofline"][k] for k in ("achieved","frac","launch_ms","sustained_launch_ms")})
e=d["extra"]
print("conv3x3", e["q8conv_3x3_56x56x64_b128"])
print("dw", e["q8dwconv_mobilenetv2_layers"])
s=e["mobilenetv2_sweep"]; print("sweep", {k:s[k] for k in s if k not in ("layers","cpu_baseline")})
print(" ".join(f"{r['layer']}:{r['ms']*1000:.1f}" for r in s["layers"]))
print("net", e["mobilenetv2_network"]["images_per_s"], e["mobilenetv2_network_fused"]["images_per_s"])
for k,v in e["q8dwconv_5x5_dilated_and_realistic_scale"].items(): print(k, v["kernel"], v["ms"], v["gbs"])
PY
