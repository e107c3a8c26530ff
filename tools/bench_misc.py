#!/usr/bin/env python3
"""Isolated timing of the memory-bound kernels of the step at the c2 geometry, with achieved GB/s (algorithmic bytes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirror_amd import kernels as K

dev, bf, f32 = "cuda", torch.bfloat16, torch.float32
B, n, n_p, D, h = 16, 4097, 4352, 512, 8


def timeit(name, fn, nbytes, reps=10):
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
    print(f"{name:46s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:8.1f} GB/s", flush=True)


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "ln"):
    x = torch.randn(B, n, D, device=dev)
    dy = torch.randn(B, n_p, D, device=dev).to(bf)
    gamma = torch.ones(D, device=dev)
    mean = torch.zeros(B * n, device=dev)
    rstd = torch.ones(B * n, device=dev)
    dx = torch.zeros_like(x)
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    pad = n_p - n
    dyv = dy[:, pad:]
    timeit("layernorm_bwd f32 x, bf16 dy, acc_dx", lambda: K.layernorm_bwd(dyv, x, gamma, mean, rstd, dx, dg, db, B, n, D, n * D, n_p * D, True),
           B * n * D * (2 + 4 + 4 + 4))
    timeit("layernorm_bwd f32 x, bf16 dy", lambda: K.layernorm_bwd(dyv, x, gamma, mean, rstd, dx, dg, db, B, n, D, n * D, n_p * D, False),
           B * n * D * (2 + 4 + 4))
    y = torch.empty(B, n_p, D, device=dev, dtype=bf)
    beta = torch.zeros(D, device=dev)
    timeit("layernorm_fwd f32 -> bf16", lambda: K.layernorm_fwd(x, gamma, beta, y[:, pad:], mean, rstd, B, n, D, n * D, n_p * D, 1e-5),
           B * n * D * (4 + 2))
if which in ("all", "resconv"):
    qkv = torch.randn(B, n_p, 3 * D, device=dev).to(bf)
    w = torch.randn(h, 33, device=dev)
    out = torch.zeros(B, n_p, D, device=dev, dtype=bf)
    dout = torch.randn(B, n_p, D, device=dev).to(bf)
    v = qkv[..., 2 * D:]
    timeit("resconv fwd (accumulate into out)", lambda: K.resconv(v, w, out, h, transpose=False, accumulate=True), B * n_p * D * 2 * 3)
    dq = torch.zeros_like(qkv)
    timeit("resconv bwd-data (accumulate into dv)", lambda: K.resconv(dout, w, dq[..., 2 * D:], h, transpose=True, accumulate=True), B * n_p * D * 2 * 3)
    dw = torch.zeros(h * 33, device=dev)
    timeit("resconv_wgrad", lambda: K.resconv_wgrad(v, dout, dw, h), B * n_p * D * 2 * 2)
if which in ("all", "mask"):
    T = n
    dy = torch.randn(B, T, D, device=dev)
    mask = (torch.rand(B, T - 1, device=dev) > 0.5).float()
    dtok = torch.zeros(D, device=dev)
    dpos = torch.zeros(T, D, device=dev)
    timeit("mask_apply_bwd f32 [16,4097,512]", lambda: K.mask_apply_bwd(dy, mask, dtok, dpos, B, T, D, 1, False), B * T * D * 4)
    # the step's form: bf16 dx out of place + the bias column sums; four gradient buffers in turn (536 MB: past the 256 MB Infinity Cache)
    dys = [torch.randn(B, T, D, device=dev) for _ in range(4)]
    dxb = torch.empty(B, T, D, device=dev, dtype=torch.bfloat16)
    dbias = torch.zeros(D, device=dev)
    turn = [0]
    def step_form():
        turn[0] = (turn[0] + 1) % 4
        K.mask_apply_bwd(dys[turn[0]], mask, dtok, dpos, B, T, D, 1, False, out=dxb, dbias=dbias)
    timeit("mask_apply_bwd f32 -> bf16 + dbias, cold", step_form, B * T * D * 6, reps=12)
    x = torch.randn(B, T, D, device=dev)
    tok = torch.zeros(D, device=dev)
    pos = torch.zeros(T, D, device=dev)
    timeit("mask_apply_fwd f32", lambda: K.mask_apply_fwd(x, mask, tok, pos, B, T, D, 1, False), B * T * D * 8)
if which in ("all", "pinv"):
    a2 = torch.randn(B, h, 256, 256, device=dev).softmax(-1)
    timeit("pinv_absmax [128,256,256]", lambda: K.pinv_absmax(a2), a2.numel() * 4)
if which in ("all", "ppeg"):
    for Bp, S in ((B, 64), (8, 91)):          # c2: 16 x 64 x 64 tokens; config 4: 8 x 91 x 91 (8192 + 89 square-pad tokens)
        x = torch.randn(Bp, 1 + S * S, D, device=dev)
        dout = torch.randn_like(x)
        dm = torch.zeros(D * 49, device=dev)
        dbs = torch.zeros(D, device=dev)
        timeit(f"ppeg_wgrad f32 [{Bp}, {S}x{S}]", lambda: K.ppeg_wgrad(x, dout, dm, dbs, S), x.numel() * 8)
        merged = torch.randn(49 * D, device=dev)
        bsum = torch.randn(D, device=dev)
        timeit(f"ppeg_fwd f32 [{Bp}, {S}x{S}]", lambda: K.ppeg(x, merged, bsum, S, False), x.numel() * 8)
if which in ("all", "pinv"):
    a2 = torch.randn(B, h, 256, 256, device=dev).softmax(-1)
    stt = K.pinv_absmax(a2)
    z0 = K.pinv_z0(a2, stt)
    dz0 = torch.randn_like(a2)
    dxx = torch.zeros_like(a2)
    timeit("pinv_z0_bwd [128,256,256]", lambda: K.pinv_z0_bwd(a2, z0, dz0, stt, dxx), a2.numel() * 16)
