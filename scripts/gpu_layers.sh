#!/bin/bash
# measurement only: time sweep layers ($2, space separated) under each value ($3...) of env var $1
VAR=$1; LAYERS=$2; shift 2
for v in "$@"; do
  echo -n "$VAR=$v:"
  for l in $LAYERS; do
    env $VAR=$v python bench.py --layer $l 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' L%d %.1fus' % (d['layer'], d['ms']*1e3), end='')"
  done
  echo
done
