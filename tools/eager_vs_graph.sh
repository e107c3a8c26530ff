#!/bin/bash
# N=1 step time: whole-step HIP graph vs eager launch (what a multi-GPU rank runs), the latter with and without the
# graphed RNA branch (mirror_amd/graphed.py)
run() { env "$@" python bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  run MIRROR_GRAPH=1
  run MIRROR_GRAPH=0 MIRROR_RNA_GRAPH=1
  run MIRROR_GRAPH=0 MIRROR_RNA_GRAPH=0
done
