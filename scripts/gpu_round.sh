#!/bin/bash
# One gpurun call: GPU-tier tests, bench line, rocprofv3 kernel stats. Everything lands in gpurun_out/.
# usage: gpurun --timeout 1500 -- bash scripts/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== device"; rocminfo 2>/dev/null | grep -E "gfx|Compute Unit" | head -4; nproc
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider 2>&1 | tail -n 120 > $OUT/pytest_gpu.log
tail -n 45 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 5 | tee $OUT/smoke.log
echo "== bench"
timeout 600 python bench.py --steps 30 --warmup 5 2>&1 | tail -n 3 | tee $OUT/bench.json
echo "== rocprofv3 kernel stats (headline GEMM only)"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o gemm -- python $OLDPWD/bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline > $OLDPWD/$OUT/prof_run.log 2>&1)
find $OUT/prof -name "*kernel_stats*" | head -3
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -n 12 $f; done
echo "== rocprofv3 kernel stats (whole bench: conv, depthwise layers, MobileNetV2 sweep)"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_all -o all -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OLDPWD/$OUT/prof_all_run.log 2>&1)
for f in $(find $OUT/prof_all -name "*kernel_stats*.csv" | head -1); do cut -c1-160 $f | head -n 16; done
echo "== PMC: HBM-side traffic of one depthwise layer (MobileNetV2 layer 8, 56x56x144) and of the pointwise layer 4"
bash scripts/gpu_pmc_layer2.sh $TAG dw8_traffic 8 TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum
bash scripts/gpu_pmc_layer2.sh $TAG pw4_traffic 4 TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum
echo "== PMC: VALU instruction count and activity of the depthwise layer 8 kernel"
bash scripts/gpu_pmc_layer2.sh $TAG dw8_valu 8 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY | tee $OUT/pmc_dw8_valu.txt
