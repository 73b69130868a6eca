#!/bin/bash
# One gpurun call: GPU-tier tests, smoke, the bench line, rocprofv3 kernel stats (GEMM only + whole bench), PMC passes of the GEMM. gpurun_out/<tag>/.
TAG=${1:-r06c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "== device"; rocminfo 2>/dev/null | grep -E "gfx|Compute Unit" | head -4; nproc
echo "== pytest -m gpu"
timeout 1800 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider 2>&1 | tail -n 60 > $OUT/pytest_gpu.log
tail -n 12 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 5 | tee $OUT/smoke.log
echo "== bench"
timeout 900 python bench.py --steps 20 --warmup 5 --full-out $OUT/bench_full.json 2> $OUT/bench.err | tail -n 1 > $OUT/bench.json
wc -c $OUT/bench.json
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print(json.dumps({k: d[k] for k in ("value", "ms_per_step")}), json.dumps(d["roofline"], indent=0).replace("\n", " "))
f = json.load(open("$OUT/bench_full.json"))
for net, v in f["extra"].get("conv_lists", {}).items():
    print(net, v["images_per_s_by_sum_of_layers"], v["frac_of_bound"])
    for r in v["layers"]:
        print("  ", r)
print([ (r["layer"], r["kernel"], r["ms"]) for r in f["extra"]["mobilenetv2_sweep"]["layers"]])
PY
tail -n 5 $OUT/bench.err
echo "== rocprofv3 kernel stats (headline GEMM only)"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o gemm -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --full-out "" > $OLDPWD/$OUT/prof_run.log 2>&1)
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do cut -c1-200 $f | head -n 6; cp $f $OUT/rocprof_kernel_stats_gemm.csv; done
echo "== rocprofv3 kernel stats (whole bench)"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_all -o all -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --full-out "" > $OLDPWD/$OUT/prof_all_run.log 2>&1)
for f in $(find $OUT/prof_all -name "*kernel_stats*.csv" | head -1); do cut -c1-160 $f | head -n 14; cp $f $OUT/rocprof_kernel_stats_all.csv; done
rm -rf $OUT/prof $OUT/prof_all
echo "== PMC: fabric traffic and SQ counters of the GEMM (auto = 16x16x64; 20 = 32x32x32)"
for V in 0 28; do
bash scripts/gpu_pmc_cmd.sh $TAG gemm_tcc_$V "python tools/gemm_ab.py --only $V $( [ $V = 28 ] && echo --kzp 126 )" TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_EA0_WRREQ_64B_sum | tee $OUT/pmc_gemm_tcc_v$V.txt
bash scripts/gpu_pmc_cmd.sh $TAG gemm_sq_$V "python tools/gemm_ab.py --only $V $( [ $V = 28 ] && echo --kzp 126 )" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU | tee $OUT/pmc_gemm_sq_v$V.txt
bash scripts/gpu_pmc_cmd.sh $TAG gemm_lds_$V "python tools/gemm_ab.py --only $V $( [ $V = 28 ] && echo --kzp 126 )" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS | tee $OUT/pmc_gemm_lds_v$V.txt
rm -rf $OUT/pmc_gemm_tcc_$V $OUT/pmc_gemm_sq_$V $OUT/pmc_gemm_lds_$V
done
