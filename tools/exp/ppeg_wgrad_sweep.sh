#!/bin/bash
# PPEG weight-gradient tiling sweep on the GPU box (from the repo root): rebuilds ppeg.o with each (PW_X, PW_ROWS, PW_RED), checks the
# kernel tests, times it at the c2 geometry (tools/bench_misc.py ppeg).  The last variant built is the source default.
set -u
for v in "4 64 0" "2 64 1" "2 64 0" "4 32 1" "4 64 1"; do
  set -- $v
  touch mirror_amd/csrc/ppeg.hip
  make -s -C mirror_amd/csrc FLAGS_ppeg="-DPW_X=$1 -DPW_ROWS=$2 -DPW_RED=$3" > /dev/null 2>&1 || { echo "build failed $v"; exit 1; }
  echo "== PW_X=$1 PW_ROWS=$2 PW_RED=$3"
  python -m pytest tests/test_kernels_gpu.py -q -x -k ppeg 2>&1 | tail -1
  python tools/bench_misc.py ppeg 2>/dev/null | grep ppeg
done
