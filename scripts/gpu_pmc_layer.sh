#!/bin/bash
# PMC counters for individual MobileNetV2 sweep layers. usage: gpu_pmc_layer.sh <tag> "<layer list>"
TAG=$1; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for L in $2; do
  echo "=== layer $L"
  python $REPO/bench.py --layer $L --steps 20 --warmup 3 2>/dev/null | tail -n 1
  i=0
  for CTRS in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
              "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
              "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/l${L}_p$i -o pmc -- python $REPO/bench.py --layer $L --steps 5 --warmup 2 > $OUT/l${L}_p$i.log 2>&1
    f=$(find $OUT/l${L}_p$i -name "*counter_collection.csv" | head -1)
    python - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
try:
    rows = list(csv.DictReader(open(f)))
except Exception as e:
    print("no counters", e); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"]
    if "q8_" not in k: continue
    k = k.split("(")[0][-48:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print("  ", k, {c: round(v / cnt[(k, c)]) for c, v in acc[k].items()})
PY
  done
done
