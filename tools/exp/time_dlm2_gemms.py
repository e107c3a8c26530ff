#!/usr/bin/env python3
"""The two landmark-gradient products behind the chain backward (dk_l = dS2^T q_l, dq_l = dS2 k_l; f32 dS2, bf16 landmarks), alone on the chip."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mirror_amd import kernels as K
from mirror_amd._lib import MH_BF16
dev = "cuda"
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
B, h, m, dh = 16, 8, 256, 64
dS2 = torch.randn(B, h, m, m, device=dev)
lm = (torch.randn(B, m, 2 * h * dh, device=dev) * .5).to(torch.bfloat16)
ql = lm[..., :h * dh].view(B, m, h, dh).permute(0, 2, 1, 3)
kl = lm[..., h * dh:].view(B, m, h, dh).permute(0, 2, 1, 3)
out = torch.zeros(B, m, 2 * h * dh, device=dev)
ok = out[..., h * dh:].view(B, m, h, dh).permute(0, 2, 1, 3)
oq = out[..., :h * dh].view(B, m, h, dh).permute(0, 2, 1, 3)
print(f"dS2^T q_l: {t(lambda: K.gemm(dS2.transpose(-1, -2), ql, out=ok, alpha=0.125, mma=MH_BF16)):6.1f} us")
print(f"dS2   k_l: {t(lambda: K.gemm(dS2, kl, out=oq, alpha=0.125, mma=MH_BF16)):6.1f} us")
dS2b = dS2.to(torch.bfloat16)
print(f"bf16 dS2^T q_l: {t(lambda: K.gemm(dS2b.transpose(-1, -2), ql, out=ok, alpha=0.125, mma=MH_BF16)):6.1f} us")
print(f"bf16 dS2   k_l: {t(lambda: K.gemm(dS2b, kl, out=oq, alpha=0.125, mma=MH_BF16)):6.1f} us")
