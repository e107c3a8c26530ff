#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "pinv" > gpurun_out/r5f_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r5f_tests.log
[ $rc -ne 0 ] && exit 1
for i in 1 2 3; do
  echo "== current"; python tools/bench_chain.py 2>&1 | grep -E "pinv_chain_(fwd|bwd)"
  echo "== base (round 4)"; (cd _ab_base/base && PYTHONPATH=$PWD python tools/bench_chain.py 2>&1 | grep -E "pinv_chain_(fwd|bwd)")
done | tee gpurun_out/r5f_chain.txt
