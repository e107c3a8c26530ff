#!/usr/bin/env python3
"""fp8 (e4m3, f8f6f4 MFMA) vs bf16 forward projection at the step's shapes: GEMM alone and with the quantisation passes."""
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
    return e0.elapsed_time(e1) / reps * 1e3


for M, N, Kd in ((69632, 1536, 512), (65536, 512, 1024), (69632, 512, 512)):
    x = (torch.randn(M, Kd, device=dev) * .5).to(bf)
    w = (torch.randn(N, Kd, device=dev) * .5).to(bf)
    out = torch.empty(M, N, device=dev, dtype=bf)
    xq, sx = K.quant_fp8(x)
    wq, sw = K.quant_fp8(w)
    fl = 2.0 * M * N * Kd / 1e6
    us_b = t(lambda: K.gemm(x, w.t(), out=out, mma=MH_BF16))
    us_f = t(lambda: K.gemm_fp8(xq, sx, wq, sw, out))
    us_q = t(lambda: K.quant_fp8(x))
    print(f"[{M},{N},{Kd}]  bf16 {us_b:7.1f} us {fl / us_b:7.1f} TF/s | fp8 gemm {us_f:7.1f} us {fl / us_f:7.1f} TF/s | quantise x {us_q:6.1f} us", flush=True)
