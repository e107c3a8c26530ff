#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout -k 10 1000 python -m pytest tests/test_fused_epilogue_gpu.py tests/test_engine_gpu.py tests/test_model_gpu.py tests/test_bench_path_gpu.py -x -q -k "round5 or graph_replay or landmark or lm or train_mode or engine_step or bf16 or c2 or c4 or template" > gpurun_out/r5ac_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r5ac_tests.log
[ $rc -ne 0 ] && exit 1
bash tools/exp/ab_flags_n.sh 6 functional._DEFER_QK=False 2>&1 | tee gpurun_out/r5ac_defer_qk_ab.txt
for v in "" "--off _DEFER_QK"; do echo "== probes $v"; MIRROR_PROBE=1 python3 tools/exp/probe_timeline.py $v 2>&1 | grep -v amdgpu | awk 'NR>1{printf "%s=%s ", $NF, $1}' | tr ' ' '\n' | grep -E "fc1_out.fwd|wsi_enc_out.fwd|decoder_end|loss_done|rna_enc_out.bwd|wsi_enc_out.bwd|fc1_out.bwd|adam" | tr '\n' ' '; echo; done | tee gpurun_out/r5ac_probes.txt
