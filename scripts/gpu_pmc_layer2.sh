#!/bin/bash
# one rocprofv3 --pmc pass over one sweep layer. usage: gpu_pmc_layer2.sh <tag> <pass-name> <layer> COUNTER...
TAG=$1; NAME=$2; LAYER=$3; shift 3
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout -s KILL 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$NAME -o pmc -- python $REPO/bench.py --layer $LAYER --dw-kernel ${DWK:-0} > $OUT/pmc_$NAME.log 2>&1
f=$(find $OUT/pmc_$NAME -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:50]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    if "qnnp" in k or "q8_" in k:
        print(k, {c: round(v / cnt[(k, c)]) for c, v in acc[k].items()})
PY
