#!/usr/bin/env python3
"""nys_sim2 / nys_dz_dav alone on the chip (c2 shapes: B = 16, h = 8, m = 256, dh = 64): us per launch from a graph of 20 launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mirror_amd import kernels as K
B, h, m, D = 16, 8, 256, 512
lm = (torch.randn(B, m, 2 * D, device="cuda") * 0.5).to(torch.bfloat16)
st = torch.zeros(4, device="cuda").view(torch.int64)
def run():
    st.zero_()
    return K.nys_sim2(lm, h, 64 ** -0.5, st)
for _ in range(3):
    run()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20):
        run()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print(f"nys_sim2 (+ an 8-byte fill): {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per launch")
