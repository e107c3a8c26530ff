#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_fused_epilogue_gpu.py tests/test_engine_gpu.py -x -q -k "colsum or mask_apply or masked_mse or layernorm or round5 or fanout or retention or graph_replay or loss" > gpurun_out/r5u_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r5u_tests.log
[ $rc -ne 0 ] && exit 1
python tools/exp/ab_dirs.py --rounds 4 --steps 30 _ab_base/prev . 2>&1 | tee gpurun_out/r5u_ab_vs_prev.txt
bash tools/trace_step_raw.sh r5v > /dev/null 2>&1 && python tools/prof_step_listing.py gpurun_out/r5v_kernel_trace.csv > gpurun_out/r5v_step_listing.txt 2>&1; rm -f gpurun_out/r5v_kernel_trace.csv
grep -n "mse_masked\|mask_apply_bwd_vec\|layernorm_bwd_ws\|colsum\|fanout" gpurun_out/r5v_step_listing.txt | cut -c1-110
