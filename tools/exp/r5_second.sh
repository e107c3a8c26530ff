#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
python -m pytest tests/test_kernels_gpu.py -x -q -k "resconv" > gpurun_out/r5b_tests.log 2>&1; echo "tests rc=$?"
tail -2 gpurun_out/r5b_tests.log
for i in 1 2; do
  for w in 1 0; do echo "MH_CHAIN_WPF=$w"; MH_CHAIN_WPF=$w python tools/bench_chain.py 2>&1 | grep -E "pinv_chain_(fwd|bwd)|Error|error"; done
done | tee gpurun_out/r5b_chain_wpf.txt
python tools/bench_nys.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5b_bench_nys.txt
python tools/exp/ab_dirs.py --rounds 4 --steps 30 _ab_base/base . 2>&1 | tee gpurun_out/r5b_ab_resconv_fused.txt
