#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 1000 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_model_gpu.py -x -q -k "noise or graph or engine or mask or eval or validate or prototype" > gpurun_out/r5z_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r5z_tests.log
[ $rc -ne 0 ] && exit 1
bash tools/exp/ab_flags_n.sh 6 models.mirror._OWN_NOISE=False 2>&1 | tee gpurun_out/r5z_own_noise_ab.txt
