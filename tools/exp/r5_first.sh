#!/bin/bash
# round 5, first GPU call: new fused res_conv kernels (tests + isolated timing), the chain backward without the in-loop spill, a bench line
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "resconv or nys_ or split_k or pinv or large_tile" > gpurun_out/r5a_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r5a_tests.log
tail -5 gpurun_out/r5a_tests.log
for i in 1 2; do
  for w in 1 0; do echo "MH_CHAIN_WPF=$w"; MH_CHAIN_WPF=$w python tools/bench_chain.py 2>&1 | grep -E "pinv_chain_(fwd|bwd)"; done
done | tee gpurun_out/r5a_chain_wpf.txt
python tools/bench_nys.py 2>&1 | tee gpurun_out/r5a_bench_nys.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r5a_bench.err | tee gpurun_out/r5a_bench.json
