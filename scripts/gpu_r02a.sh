#!/bin/bash
# round-2 first GPU call: new tests, full GPU tier, smoke, bench line, rocprofv3 stats of the GEMM.
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== device"; rocminfo 2>/dev/null | grep -E "gfx|Compute Unit" | head -4; nproc
echo "== new tests first"
QNNP_WRITE_SWEEP_KERNELS=1 timeout 600 python -m pytest tests/test_gpu_sweep_bench_batch.py -q -p no:cacheprovider 2>&1 | tail -n 15
cp tests/golden/sweep_kernels.json $OUT/ 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_multidevice.py tests/test_gpu_c_consumer.py -q -p no:cacheprovider 2>&1 | tail -n 30
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider 2>&1 | tail -n 120 > $OUT/pytest_gpu.log
tail -n 40 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 5 | tee $OUT/smoke.log
echo "== bench"
timeout 600 python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tail -n 1 > $OUT/bench.json
tail -n 5 $OUT/bench.err
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read())
print({k:d[k] for k in ("value","ms_per_step","n_gpus")}, d["roofline"])
e=d["extra"]
print("conv3x3", e["q8conv_3x3_56x56x64_b128"])
print("dw", e["q8dwconv_mobilenetv2_layers"])
s=e["mobilenetv2_sweep"]; print("sweep", {k:s[k] for k in s if k not in ("layers","cpu_baseline")})
for r in s["layers"]: print(r)
print("net", e["mobilenetv2_network"]["images_per_s"], e["mobilenetv2_network_fused"]["images_per_s"])
print(e["q8dwconv_5x5_dilated_and_realistic_scale"])
print(e["next_rows"])
print(d["cpu_baseline"])
PY
echo "== bench --gpus 2 on a 1-GPU box must fail loudly"
timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 --no-extra --no-cpu-baseline; echo "rc=$?"
echo "== rocprofv3 kernel stats (headline GEMM only)"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o gemm -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $OLDPWD/$OUT/prof_run.log 2>&1)
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -n 6 $f | cut -c1-200; done
tail -n 1 $OUT/prof_run.log | cut -c1-600
