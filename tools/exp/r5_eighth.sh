#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_fused_epilogue_gpu.py -x -q -k "gemm or epilogue or retention or to_out" > gpurun_out/r5n_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r5n_tests.log
[ $rc -ne 0 ] && exit 1
python tools/exp/time_epi_gemms.py 2>&1 | grep -v amdgpu | tee gpurun_out/r5n_epi_gemms.txt
python tools/exp/ab_dirs.py --rounds 4 --steps 30 _ab_base/prev . 2>&1 | tee gpurun_out/r5n_ab_vs_prev.txt
