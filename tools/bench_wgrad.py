#!/usr/bin/env python3
"""Weight-gradient GEMM dW[N, K] += dy^T x (bf16 operands, f32 split-K atomics) vs the split factor."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirror_amd import kernels as K
from mirror_amd._lib import MH_BF16
dev, bf = "cuda", torch.bfloat16


def run(N, Kd, rows, split):
    dy = (torch.randn(rows, N, device=dev) * .5).to(bf)
    x = (torch.randn(rows, Kd, device=dev) * .5).to(bf)
    dw = torch.zeros(N, Kd, device=dev)
    fn = lambda: K.gemm(dy.t(), x, out=dw, accumulate=True, split_k=split, mma=MH_BF16)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"dW[{N:5d},{Kd:5d}] rows={rows:6d} split={split:3d}  {ms * 1e3:8.1f} us  {2.0 * rows * N * Kd / ms / 1e9:8.1f} TF/s", flush=True)


for N, Kd, rows in ((512, 512, 65552), (1536, 512, 69632), (512, 1024, 65536), (512, 512, 69632)):
    for split in (4, 8, 16, 21, 32, 64, 128):
        run(N, Kd, rows, split)
