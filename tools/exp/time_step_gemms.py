#!/usr/bin/env python3
"""A few of the step's 256 x 256-tile products, each alone on the chip: median of 5 batches of 40 launches (run once per library
build with MIRROR_HIP_LIB=... to compare kernels on one box)."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mirror_amd import kernels as K
from mirror_amd import functional as Fn
from mirror_amd._lib import MH_BF16
dev, bf, f32 = "cuda", torch.bfloat16, torch.float32
def t(fn, reps=40, batches=5):
    for _ in range(5): fn()
    r = []
    for _ in range(batches):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(r), min(r)
rnd = lambda *s: (torch.randn(*s, device=dev) * .5).to(bf)
M = 65536
cases = []
a5, w1 = rnd(M, 512), rnd(1024, 512); o1 = torch.empty(M, 1024, device=dev, dtype=bf)
cases.append(("q|k   [65536 x 512] x [512 x 1024] -> bf16", lambda: K.gemm(a5, w1.t(), out=o1, mma=MH_BF16)))
w2 = rnd(512, 512); o2 = torch.empty(M, 512, device=dev, dtype=f32)
cases.append(("plain [65536 x 512] x [512 x 512] -> f32", lambda: K.gemm(a5, w2.t(), out=o2, mma=MH_BF16)))
a10, w3 = rnd(M, 1024), rnd(512, 1024); b3 = torch.randn(512, device=dev)
cases.append(("bias  [65536 x 1024] x [1024 x 512] + b -> f32", lambda: K.gemm(a10, w3.t(), out=o2, bias=b3, mma=MH_BF16)))
cases.append(("relu  [65536 x 1024] x [1024 x 512] + b, ReLU -> f32", lambda: K.gemm(a10, w3.t(), out=o2, bias=b3, act=K.ACT_RELU, mma=MH_BF16)))
dy5 = rnd(M, 512); dw = torch.zeros(512, 1024, device=dev, dtype=f32)
sk = Fn._split_k_for(M, 512, 1024)
cases.append((f"wgrad [512 x 65536] x [65536 x 1024] += f32, split {sk}", lambda: K.gemm(dy5.t(), a10, out=dw, accumulate=True, split_k=sk, mma=MH_BF16)))
dy15 = rnd(M + 4096, 1536); x5 = rnd(M + 4096, 512); dw2 = torch.zeros(1536, 512, device=dev, dtype=f32)
sk2 = Fn._split_k_for(M + 4096, 1536, 512)
cases.append((f"wgrad [1536 x 69632] x [69632 x 512] += f32, split {sk2}", lambda: K.gemm(dy15.t(), x5, out=dw2, accumulate=True, split_k=sk2, mma=MH_BF16)))
o5 = torch.empty(M, 512, device=dev, dtype=bf)
cases.append(("dgrad [65536 x 1536] x [1536 x 512] -> bf16", lambda: K.gemm(dy15[:M], rnd(1536, 512), out=o5, mma=MH_BF16)))
for name, fn in cases:
    med, mn = t(fn)
    print(f"{name:62s} {med:8.1f} us (min {mn:.1f})", flush=True)
