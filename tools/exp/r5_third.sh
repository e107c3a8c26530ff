#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
python -m pytest tests/test_kernels_gpu.py -x -q -k "resconv or nys_" > gpurun_out/r5c_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r5c_tests.log
python tools/bench_nys.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5c_bench_nys.txt
python tools/exp/ab_dirs.py --rounds 3 --steps 30 _ab_base/base . 2>&1 | tee gpurun_out/r5c_ab_all.txt
python tools/exp/ab_stat.py --rounds 3 - MIRROR_A1_DQ_WINDOW=0 MIRROR_TO_OUT_WGRAD_WINDOW=0 MIRROR_A1_DQ_WINDOW=0+MIRROR_TO_OUT_WGRAD_WINDOW=0 2>&1 | tee gpurun_out/r5c_ab_switches.txt
