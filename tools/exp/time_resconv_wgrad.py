#!/usr/bin/env python3
"""res_conv weight gradient alone on the chip (c2: B = 16, n_p = 4352, 8 heads x 64)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mirror_amd import kernels as K
B, n_p, D, h = 16, 4352, 512, 8
qkv = torch.randn(B, n_p, 3 * D, device="cuda").to(torch.bfloat16)
dout = torch.randn(B, n_p, D, device="cuda").to(torch.bfloat16)
dw = torch.zeros(h * 33, device="cuda")
def run(): K.resconv_wgrad(qkv[..., 2 * D:], dout, dw, h)
for _ in range(5): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): run()
torch.cuda.synchronize(); print(f"resconv_wgrad: {(time.perf_counter() - t0) / 100 * 1e6:.1f} us")
