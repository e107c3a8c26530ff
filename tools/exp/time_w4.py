#!/usr/bin/env python3
"""The four-wave 256 x 256 tile kernel (mh_gemm_w4) against the persistent 8-wave kernel behind mh_gemm on the step's forward shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mirror_amd import kernels as K
from mirror_amd._lib import MH_BF16
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)


def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, N, Kd in ((65536, 1024, 512), (65536, 512, 512), (65536, 1536, 512), (65536, 512, 1024), (65536, 512, 1536), (4096, 512, 512)):
    a = torch.randn(M, Kd, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, Kd, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device=dev, generator=g)
    ref = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    K.gemm(a, w.t(), out=ref, bias=b, mma=MH_BF16)
    got = K.gemm_w4(a, w, b)
    err = float((got.float() - ref.float()).abs().max())
    exact = bool(torch.equal(got, ref))
    ok = exact or err <= 2e-2 * float(ref.float().abs().max())
    t_ref = timeit(lambda: K.gemm(a, w.t(), out=ref, bias=b, mma=MH_BF16))
    t_w4 = timeit(lambda: K.gemm_w4(a, w, b, out=got))
    fl = 2.0 * M * N * Kd
    print(f"[{M} x {Kd}] x [{Kd} x {N}]: mh_gemm {t_ref:7.1f} us ({fl / t_ref / 1e6:6.0f} TF/s)   w4 {t_w4:7.1f} us ({fl / t_w4 / 1e6:6.0f} TF/s)   "
          f"{'bit-equal' if exact else f'max err {err:.3e}'} {'OK' if ok else 'MISMATCH'}", flush=True)
