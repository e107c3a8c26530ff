#!/bin/bash
# usage: tools/ab_env.sh VAR v1 v2 ... : default bench line per value of an environment switch, interleaved twice
var=$1; shift
for rep in 1 2; do for v in "$@"; do
  env $var=$v python bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v', d['value'], d['ms_per_step'])"
done; done
