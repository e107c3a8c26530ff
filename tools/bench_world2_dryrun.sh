#!/bin/bash
# Dry run of bench.py's multi-rank branch on a one-GPU box: two ranks share GPU 0, gloo instead of RCCL (numbers are
# meaningless — gloo stages every bucket through the host — the point is that the N > 1 path runs end to end and prints its line).
MIRROR_BENCH_DIST=gloo:shared python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 4 --warmup 3 --batch 4 --no-cpu-baseline 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -3
