#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "nys_ or pinv_chain" > gpurun_out/r5e_tests.log 2>&1; rc=$?; echo "tests rc=$rc"
tail -6 gpurun_out/r5e_tests.log
[ $rc -ne 0 ] && exit 1
python tools/bench_nys.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5e_bench_nys.txt
python tools/exp/ab_stat.py --rounds 3 - MIRROR_A3_BWD_ONE_PASS=0 2>&1 | tee gpurun_out/r5e_ab_a3_one_pass.txt
python tools/exp/ab_dirs.py --rounds 3 --steps 30 _ab_base/base . 2>&1 | tee gpurun_out/r5e_ab_all.txt
