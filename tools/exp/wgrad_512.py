#!/usr/bin/env python3
"""Where does the 512 x 512 weight gradient over 16 x 4097 rows (to_out, layer 2 / 3) spend its 107 us?"""
import torch
from mirror_amd import kernels as K
from mirror_amd._lib import MH_BF16

dev, bf = "cuda", torch.bfloat16


def timeit(name, fn, flops, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:60s} {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TF/s", flush=True)


B, D = 16, 512
for rows in (4096, 4097):
    x = torch.randn(B, 4352, D, device=dev).to(bf)[:, -rows:]       # row window of the padded buffer
    dy = torch.randn(B, rows, D, device=dev).to(bf)
    dw = torch.zeros(D, D, device=dev)
    fl = 2.0 * B * rows * D * D
    for sk in (1, 2, 4, 8):
        timeit(f"batched rows={rows} split_k={sk}", lambda: K.gemm(dy.transpose(-1, -2), x, out=dw.expand(B, D, D), accumulate=True, split_k=sk, mma=MH_BF16), fl)
    xf = x.contiguous().reshape(-1, D)
    dyf = dy.reshape(-1, D)
    for sk in (16, 32, 64, 128):
        timeit(f"flat rows={B * rows} split_k={sk}", lambda: K.gemm(dyf.t(), xf, out=dw, accumulate=True, split_k=sk, mma=MH_BF16), fl)

print("--- pieces of the rows = 4097 batched call")
rows = 4097
x = torch.randn(B, 4352, D, device=dev).to(bf)[:, -rows:]
dy = torch.randn(B, rows, D, device=dev).to(bf)
dw = torch.zeros(D, D, device=dev)
timeit("main part only (same pointers / strides, K = 4096)", lambda: K.gemm(dy[:, :4096].transpose(-1, -2), x[:, :4096], out=dw.expand(B, D, D), accumulate=True, split_k=4, mma=MH_BF16), fl)
timeit("tail only (K = 1 per batch)", lambda: K.gemm(dy[:, 4096:].transpose(-1, -2), x[:, 4096:], out=dw.expand(B, D, D), accumulate=True, split_k=1, mma=MH_BF16), 1.0)
xf = x.contiguous().reshape(-1, D)
dyf = dy.reshape(-1, D)
timeit("flat main only (K = 65536)", lambda: K.gemm(dyf[:65536].t(), xf[:65536], out=dw, accumulate=True, split_k=64, mma=MH_BF16), fl)
timeit("flat tail only (K = 16)", lambda: K.gemm(dyf[65536:].t(), xf[65536:], out=dw, accumulate=True, split_k=1, mma=MH_BF16), 1.0)
