#!/usr/bin/env python3
"""Template geometry (dh = 96, m = 384, n_p = 2304, B h = 128): the batched similarity / gradient GEMMs of the composed Nystrom
path, K = 96 (ragged: 1.5 K-tiles) against the same problem padded to K = 128."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirror_amd import kernels as K
from mirror_amd._lib import MH_BF16
dev, bf, f32 = "cuda", torch.bfloat16, torch.float32
B, h, n_p, m = 16, 8, 2304, 384

def timeit(name, fn, flops, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:56s} {ms*1e3:9.1f} us  {flops/ms/1e9:8.1f} TF/s")

for dh in (96, 128):
    D = h * dh
    dout = torch.randn(B, n_p, D, device=dev).to(bf)
    dO = dout.view(B, n_p, h, dh).permute(0, 2, 1, 3)
    w2 = torch.randn(B, h, m, dh, device=dev).to(bf)
    a1 = torch.randn(B, h, n_p, m, device=dev).to(bf)
    fl = 2.0 * B * h * n_p * m * dh
    timeit(f"dS1 = dO.w2^T  [n_p x {dh}]x[{dh} x m] -> bf16", lambda: K.gemm(dO, w2.transpose(-1, -2), mma=MH_BF16), fl)
    timeit(f"dS1 -> f32", lambda: K.gemm(dO, w2.transpose(-1, -2), mma=MH_BF16, out_dtype=f32), fl)
    out = torch.empty(B, n_p, D, device=dev, dtype=bf)
    timeit(f"out = a1.w2    [n_p x m]x[m x {dh}] -> bf16 cols", lambda: K.gemm(a1, w2, out=out.view(B, n_p, h, dh).permute(0, 2, 1, 3), mma=MH_BF16), fl)
    timeit(f"dW2 = a1^T.dO  [m x n_p]x[n_p x {dh}] -> f32", lambda: K.gemm(a1.transpose(-1, -2), dO, mma=MH_BF16, out_dtype=f32), fl)
