#!/usr/bin/env python3
"""Row remainders of the big-tile split (4096 rows): 128 x 128 tiles (128 workgroups) vs 128 x 64 (256) -- MH_GEMM_HALF_TILES=0/1."""
import torch
from mirror_amd import kernels as K
from mirror_amd._lib import MH_BF16

dev, bf = "cuda", torch.bfloat16


def timeit(name, fn, flops, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:50s} {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TF/s", flush=True)


for M, N, Kd, bt in ((4096, 512, 1536, False), (4096, 512, 512, True), (4096, 512, 512, False), (4096, 1024, 512, True)):
    a = torch.randn(M, Kd, device=dev).to(bf)
    b = (torch.randn(N, Kd, device=dev).to(bf).t() if bt else torch.randn(Kd, N, device=dev).to(bf))
    out = torch.empty(M, N, device=dev, dtype=bf)
    timeit(f"{M} x {N} x {Kd} b_kc={bt}", lambda: K.gemm(a, b, out=out, mma=MH_BF16), 2.0 * M * N * Kd)
