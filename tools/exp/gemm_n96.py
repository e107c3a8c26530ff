#!/usr/bin/env python3
"""The N = 96 batched product of the template attention (attn1 @ w2: [2304 x 384] x [384 x 96] per (b, h)), alone, for counters."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mirror_amd import kernels as K
from mirror_amd._lib import MH_BF16
bf = torch.bfloat16
a = torch.randn(16, 8, 2304, 384, device="cuda").to(bf)
w = torch.randn(16, 8, 384, 96, device="cuda").to(bf)
which = sys.argv[1] if len(sys.argv) > 1 else "n96"
if which == "n128":
    w = torch.randn(16, 8, 384, 128, device="cuda").to(bf)
for _ in range(5):
    out = K.gemm(a, w, mma=MH_BF16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    out = K.gemm(a, w, mma=MH_BF16)
e1.record(); torch.cuda.synchronize()
print(which, "us per call", e0.elapsed_time(e1) * 100)
