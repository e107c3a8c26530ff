#!/usr/bin/env python3
"""bf16 GEMM rate vs K for a long thin problem and for squares (large-tile kernel unless MH_GEMM_BIG=0)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirror_amd import kernels as K
from mirror_amd._lib import MH_BF16
dev, bf = "cuda", torch.bfloat16


def run(M, N, Kd, nt=True):
    a = (torch.randn(M, Kd, device=dev) * .5).to(bf)
    b = (torch.randn(N, Kd, device=dev) * .5).to(bf) if nt else (torch.randn(Kd, N, device=dev) * .5).to(bf)
    out = torch.empty(M, N, device=dev, dtype=bf)
    fn = lambda: K.gemm(a, b.t() if nt else b, out=out, mma=MH_BF16)
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
    print(f"M={M:6d} N={N:5d} K={Kd:5d} {'NT' if nt else 'NN'}  {ms * 1e3:8.1f} us  {2.0 * M * N * Kd / ms / 1e9:8.1f} TF/s", flush=True)


for Kd in (512, 1024, 2048, 4096):
    run(69632, 1536, Kd)
run(4096, 4096, 4096)
run(8192, 8192, 8192)
run(8192, 8192, 8192, nt=False)
