#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 600 python -m pytest tests/test_fused_epilogue_gpu.py tests/test_kernels_gpu.py -x -q -k "round5 or retention_embed or layernorm" > gpurun_out/r5m_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r5m_tests.log
[ $rc -ne 0 ] && exit 1
python tools/exp/time_epi_gemms.py 2>&1 | grep -E "MASKPOS|plain  ... \+ bias -> f32" | tee gpurun_out/r5m_maskpos.txt
python tools/exp/ab_dirs.py --rounds 4 --steps 30 _ab_base/prev . 2>&1 | tee gpurun_out/r5m_ab_vs_prev.txt
