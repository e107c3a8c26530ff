#!/bin/bash
# PPEG forward / data-gradient form sweep on the GPU box (from the repo root): rebuilds ppeg.o with each flag set, checks the
# kernel tests, times it at the c2 geometry (tools/bench_misc.py ppeg).  The last variant built is the source default.
set -u
for v in "-DP2_FORM=0" "-DP2_FORM=1 -DPR_T=16" "-DP2_FORM=1 -DPR_T=64" "-DP2_FORM=1 -DPR_T=32"; do
  touch mirror_amd/csrc/ppeg.hip
  make -s -C mirror_amd/csrc FLAGS_ppeg="$v" > /dev/null 2>&1 || { echo "build failed $v"; exit 1; }
  echo "== $v"
  python -m pytest tests/test_kernels_gpu.py -q -x -k ppeg 2>&1 | tail -1
  python tools/bench_misc.py ppeg 2>/dev/null | grep ppeg
done
