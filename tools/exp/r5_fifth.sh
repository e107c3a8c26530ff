#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -x -q -k "nys_ or key_padding or d512_graph" > gpurun_out/r5h_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r5h_tests.log
[ $rc -ne 0 ] && exit 1
python tools/bench_nys.py 2>&1 | grep -E "attn1 bwd|attn3 bwd" | tee gpurun_out/r5h_bench_nys.txt
python tools/run_c4.py --batch 8 --steps 10 2>&1 | tail -1 | tee gpurun_out/r5h_c4_b8.json
python tools/run_c4.py --batch 16 --steps 10 2>&1 | tail -1 | tee gpurun_out/r5h_c4_b16.json
python tools/exp/ab_dirs.py --rounds 3 --steps 30 _ab_base/base . 2>&1 | tee gpurun_out/r5h_ab_all.txt
