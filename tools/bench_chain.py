#!/usr/bin/env python3
"""Isolated timing of the pinv chain / fused attention kernels at the c2 geometry (B*h = 128)."""
import sys
import torch
from mirror_amd import kernels as K

dev = "cuda"
BH, m, iters = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 256, 6
x = (torch.randn(1, BH, m, m, device=dev) * 2).softmax(-1)
st = K.pinv_absmax(x)
saved = torch.zeros((iters, 4, BH, m, m), device=dev, dtype=torch.bfloat16)
z0, xp = K.pinv_chain_prep(x, st, saved[0, 0])
zfT = torch.empty((BH, m, m), device=dev, dtype=torch.bfloat16)
work = torch.empty_like(saved)
dX = torch.empty((BH, m, m), device=dev)
dz0 = torch.empty((BH, m, m), device=dev)
up = K.pinv_chain_pack(torch.randn(BH, m, m, device=dev))


def timeit(name, fn, flops, reps=10):
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
    print(f"{name:28s} {ms * 1e3:9.1f} us   {flops / ms / 1e9:8.1f} TFLOP/s", flush=True)


timeit("pinv_chain_fwd", lambda: K.pinv_chain_fwd(xp, saved, zfT, iters), iters * 4 * 2.0 * m ** 3 * BH)
timeit("pinv_chain_bwd", lambda: K.pinv_chain_bwd(xp, saved, up, work, dX, dz0, iters), iters * 8 * 2.0 * m ** 3 * BH)
timeit("pinv_chain_prep", lambda: K.pinv_chain_prep(x, st, saved[0, 0]), 1.0)
timeit("pinv_chain_pack", lambda: K.pinv_chain_pack(dX), 1.0)
timeit("pinv_absmax", lambda: K.pinv_absmax(x), 1.0)

B, h, n_p = BH // 8, 8, 4352
D = 64 * h
bf = torch.bfloat16
qkv = torch.randn((B, n_p, 3 * D), device=dev).to(bf)
lm = torch.randn((B, m, 2 * D), device=dev).to(bf)
w2 = torch.randn((B, h, m, 64), device=dev).to(bf)
out = torch.empty((B, n_p, D), device=dev, dtype=bf)
dout = torch.randn((B, n_p, D), device=dev).to(bf)
dav = torch.randn((B, h, m, 64), device=dev).to(bf)
o1 = torch.empty_like(out)
lse1 = K.nys_attn1_fwd(qkv, lm, w2, out, h, 0.125, o1=o1)
delta1 = torch.empty_like(lse1)
av, lse3 = K.nys_attn3_fwd(qkv, lm, h, 0.125)
dqkv = torch.empty_like(qkv)
dw2 = torch.zeros((B, h, m, 64), device=dev)
dlm = torch.zeros((B, m, 2 * D), device=dev)
fl = 2.0 * n_p * m * 64 * B * h
timeit("nys_attn1_fwd", lambda: K.nys_attn1_fwd(qkv, lm, w2, out, h, 0.125), 2 * fl)
timeit("nys_attn3_fwd", lambda: K.nys_attn3_fwd(qkv, lm, h, 0.125), 2 * fl)
timeit("nys_attn1_bwd (2 kernels)", lambda: K.nys_attn1_bwd(qkv, lm, w2, dout, lse1, o1, delta1, dqkv, dw2, dlm, h, 0.125), 7 * fl)
timeit("nys_attn3_bwd (2 kernels)", lambda: K.nys_attn3_bwd(qkv, lm, av, dav, lse3, dqkv, dlm, h, 0.125), 7 * fl)

# ---- does the chain really overlap with the attn3 side?  (side stream vs same stream)
side = torch.cuda.Stream()
w33 = torch.randn(h, 33, device=dev)
outc = torch.zeros(B, n_p, D, device=dev, dtype=bf)


def both(concurrent: bool):
    main = torch.cuda.current_stream()
    if concurrent:
        side.wait_stream(main)
        with torch.cuda.stream(side):
            K.pinv_chain_fwd(xp, saved, zfT, iters)
    else:
        K.pinv_chain_fwd(xp, saved, zfT, iters)
    K.nys_attn3_fwd(qkv, lm, h, 0.125)
    K.resconv(qkv[..., 2 * D:], w33, outc, h, transpose=False, accumulate=False)
    if concurrent:
        main.wait_stream(side)


timeit("chain_fwd ; attn3_fwd ; resconv (serial)", lambda: both(False), 1.0)
timeit("chain_fwd || attn3_fwd ; resconv", lambda: both(True), 1.0)


def both_bwd(concurrent: bool):
    main = torch.cuda.current_stream()
    if concurrent:
        side.wait_stream(main)
        with torch.cuda.stream(side):
            K.pinv_chain_bwd(xp, saved, up, work, dX, dz0, iters)
    else:
        K.pinv_chain_bwd(xp, saved, up, work, dX, dz0, iters)
    K.nys_attn3_bwd(qkv, lm, av, dav, lse3, dqkv, dlm, h, 0.125)
    K.resconv(dout, w33, dqkv[..., 2 * D:], h, transpose=True, accumulate=True)
    if concurrent:
        main.wait_stream(side)


timeit("chain_bwd ; attn3_bwd ; resconv^T (serial)", lambda: both_bwd(False), 1.0)
timeit("chain_bwd || attn3_bwd ; resconv^T", lambda: both_bwd(True), 1.0)
