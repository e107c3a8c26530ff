#!/bin/bash
# Dry run of bench.py's multi-rank branch on a one-GPU box: `python bench.py --gpus 2` with NO external launcher (bench.py
# starts torch.distributed.run itself), two ranks sharing GPU 0 over gloo instead of RCCL.  The numbers are meaningless —
# gloo stages every bucket through the host — the point is that the N > 1 path runs end to end and prints "n_gpus": 2.
MIRROR_BENCH_DIST=gloo:shared python3 bench.py --gpus 2 --steps 4 --warmup 3 --batch 4 --no-cpu-baseline 2>&1 \
  | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -3
