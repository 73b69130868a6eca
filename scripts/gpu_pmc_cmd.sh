#!/bin/bash
# one rocprofv3 --pmc pass (counters only, no trace domains besides the kernel trace) over an arbitrary command:
#   gpu_pmc_cmd.sh <tag> <pass-name> "<command>" COUNTER...      (run from the repo root, on the GPU box)
# prints per-kernel averages of every counter; the raw CSV stays under gpurun_out/<tag>/pmc_<pass-name>/
TAG=$1; NAME=$2; CMD=$3; shift 3
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout -s KILL 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$NAME -o pmc -- bash -c "cd $REPO && $CMD" > $OUT/pmc_$NAME.log 2>&1
f=$(find $OUT/pmc_$NAME -name "*counter_collection.csv" | head -1)
[ -z "$f" ] && { echo "no counter file for $NAME"; tail -n 5 $OUT/pmc_$NAME.log; exit 0; }
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    if "qnnp" in k or "q8_" in k:
        print(k, {c: round(v / cnt[(k, c)]) for c, v in acc[k].items()}, "launches", max(cnt[(k, c)] for c in acc[k]))
PY
