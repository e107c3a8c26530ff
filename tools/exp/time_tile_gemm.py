#!/usr/bin/env python3
"""The 192 x 384 tile kernel on the template's chain products ([16, 8, 384, K] x [16, 8, K, 384]): fixed cost vs K-tile cost."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mirror_amd import kernels as K
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
    return statistics.median(r)
rnd = lambda *s: (torch.randn(*s, device=dev) * .1).to(bf)
for Kd in (64, 128, 192, 384, 768):
    a, b = rnd(16, 8, 384, Kd), rnd(16, 8, Kd, 384)
    ob = torch.empty(16, 8, 384, 384, device=dev, dtype=bf); of = torch.zeros(16, 8, 384, 384, device=dev, dtype=f32)
    K._prof = None
    us1 = t(lambda: K.gemm(a, b, out=ob, mma=MH_BF16))
    us2 = t(lambda: K.gemm(a, b.transpose(-1, -2).contiguous().transpose(-1, -2), out=ob, alpha=0.25, mma=MH_BF16)) if False else 0
    us3 = t(lambda: K.gemm(a.transpose(-1, -2).contiguous().transpose(-1, -2), b, out=of, accumulate=True, mma=MH_BF16))
    print(f"K = {Kd:4d}: bf16 out {us1:6.1f} us   f32 accumulate (A K-strided) {us3:6.1f} us   [{2*128*384*384*Kd/us1/1e6:5.0f} TF/s]", flush=True)
