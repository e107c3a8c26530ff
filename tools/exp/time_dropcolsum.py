#!/usr/bin/env python3
"""dropout backward (+ bias column sums) of a [65552, 512] f32 gradient: separate launches vs the fused one, alone on the chip."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mirror_amd import kernels as K
x = torch.randn(16, 4097, 512, device="cuda")
gb = torch.empty(x.shape, device="cuda", dtype=torch.bfloat16)
db = torch.zeros(512, device="cuda")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
def sep():
    K.dropout_lite(x, 0.1, 1, 0, None, out=gb); K.colsum(gb.view(-1, 512), db)
print(f"dropout_lite + colsum: {t(sep):.1f} us;  dropout_lite alone: {t(lambda: K.dropout_lite(x, 0.1, 1, 0, None, out=gb)):.1f} us;  fused: {t(lambda: K.dropout_lite_colsum(x, 0.1, 1, 0, None, gb, db)):.1f} us")
