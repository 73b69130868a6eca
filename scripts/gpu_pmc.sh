#!/bin/bash
# PMC counters for the headline GEMM kernel. usage: gpu_pmc.sh <tag> [ablate]
TAG=$1; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export QNNP_GFX950_ABLATE=${2:-0}
REPO=$PWD
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "(SQ|GRBM|TCC|TCP)_[A-Z0-9_]+" | sort -u > $OUT/counters_available.txt
wc -l $OUT/counters_available.txt
run() { # name, counters...
  n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$n -o pmc -- python $REPO/bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $OUT/pmc_$n.log 2>&1
  f=$(find $OUT/pmc_$n -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    if "gemm" in k or "igemm" in k:
        print(k, {c: round(v / cnt[(k, c)]) for c, v in acc[k].items()})
PY
}
run a SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run b SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
run c SQ_INST_CYCLES_VMEM_RD SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU
