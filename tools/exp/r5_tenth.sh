#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -x -q -k "adam or graph_replay or engine_step or accum or clip" > gpurun_out/r5y_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r5y_tests.log
[ $rc -ne 0 ] && exit 1
bash tools/exp/ab_flags3.sh engine._EARLY_ADAM=False 2>&1 | tee gpurun_out/r5y_flags.txt
MIRROR_PROBE=1 python3 tools/exp/probe_timeline.py 2>&1 | grep -v amdgpu > gpurun_out/r5y_probe.txt; tail -8 gpurun_out/r5y_probe.txt
