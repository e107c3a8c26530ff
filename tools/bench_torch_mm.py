#!/usr/bin/env python3
"""Library (hipBLASLt via torch) bf16 GEMM rate at the step's plain-GEMM shapes, next to this build's kernels."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirror_amd import kernels as K
from mirror_amd._lib import MH_BF16
dev, bf = "cuda", torch.bfloat16


def t(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def fwd(M, N, Kd):
    a = (torch.randn(M, Kd, device=dev) * .5).to(bf)
    w = (torch.randn(N, Kd, device=dev) * .5).to(bf)
    out = torch.empty(M, N, device=dev, dtype=bf)
    ms_l = t(lambda: torch.mm(a, w.t(), out=out))
    ms_o = t(lambda: K.gemm(a, w.t(), out=out, mma=MH_BF16))
    fl = 2.0 * M * N * Kd / 1e9
    print(f"fwd   y[{M},{N}] = x[{M},{Kd}] W^T      lib {ms_l * 1e3:7.1f} us {fl / ms_l:7.1f} TF/s | ours {ms_o * 1e3:7.1f} us {fl / ms_o:7.1f} TF/s", flush=True)


def dgrad(M, N, Kd):
    dy = (torch.randn(M, N, device=dev) * .5).to(bf)
    w = (torch.randn(N, Kd, device=dev) * .5).to(bf)
    out = torch.empty(M, Kd, device=dev, dtype=bf)
    ms_l = t(lambda: torch.mm(dy, w, out=out))
    ms_o = t(lambda: K.gemm(dy, w, out=out, mma=MH_BF16))
    fl = 2.0 * M * N * Kd / 1e9
    print(f"dgrad dx[{M},{Kd}] = dy[{M},{N}] W       lib {ms_l * 1e3:7.1f} us {fl / ms_l:7.1f} TF/s | ours {ms_o * 1e3:7.1f} us {fl / ms_o:7.1f} TF/s", flush=True)


def wgrad(M, N, Kd, split):
    dy = (torch.randn(M, N, device=dev) * .5).to(bf)
    x = (torch.randn(M, Kd, device=dev) * .5).to(bf)
    dwb = torch.empty(N, Kd, device=dev, dtype=bf)
    dw = torch.zeros(N, Kd, device=dev)
    ms_l = t(lambda: torch.mm(dy.t(), x, out=dwb))
    ms_o = t(lambda: K.gemm(dy.t(), x, out=dw, accumulate=True, split_k=split, mma=MH_BF16))
    fl = 2.0 * M * N * Kd / 1e9
    print(f"wgrad dW[{N},{Kd}] = dy^T x (rows {M})   lib {ms_l * 1e3:7.1f} us {fl / ms_l:7.1f} TF/s | ours {ms_o * 1e3:7.1f} us {fl / ms_o:7.1f} TF/s", flush=True)


fwd(69632, 1536, 512)
fwd(65536, 512, 1024)
fwd(65552, 512, 512)
dgrad(69632, 1536, 512)
dgrad(65552, 512, 512)
wgrad(69632, 1536, 512, 21)
wgrad(65536, 512, 1024, 8)
wgrad(65552, 512, 512, 16)
fwd(8192, 8192, 8192)
