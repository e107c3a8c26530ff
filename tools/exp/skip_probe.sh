#!/bin/bash
# What do the "hidden" side-stream branches cost the step in chip throughput?  EXP library (never shipped; results garbage): the named
# entry points return without launching.  bash tools/exp/skip_probe.sh <exp .so> "<entry,entry>" ...   (GPU box, repo root)
R=$PWD; LIB=$1; shift
cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R MIRROR_HIP_LIB=$LIB
run() { echo -n "$1: "; env $1 python3 $R/tools/exp/step_time.py 30 2>/dev/null | tail -1; }
for i in 1 2 3; do
  run X=0
  for s in "$@"; do run MH_EXP_SKIP=$s; done
  run X=0
done
