#!/usr/bin/env python3
"""The K = 512 projection shapes of the step, alone on the chip (run with MIRROR_HIP_LIB=<experiment build> to split their time)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mirror_amd import kernels as K
from mirror_amd._lib import MH_BF16
dev, bf = "cuda", torch.bfloat16
def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for M, N, Kd in ((65536, 1024, 512), (65536, 512, 512), (65536, 512, 1536), (65536, 2048, 512)):
    a = (torch.randn(M, Kd, device=dev) * .5).to(bf); w = (torch.randn(N, Kd, device=dev) * .5).to(bf)
    out = torch.empty(M, N, device=dev, dtype=bf)
    us = t(lambda: K.gemm(a, w.t(), out=out, mma=MH_BF16))
    print(f"[{M} x {Kd}] x [{Kd} x {N}] -> bf16: {us:7.1f} us  {2.0*M*N*Kd/us/1e6:6.0f} TF/s  ({M//256*(N//256)/256:.2f} rounds, {us/(M//256*(N//256)/256):.1f} us per round)", flush=True)
