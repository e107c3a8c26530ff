#!/usr/bin/env python3
"""GPU micro-benchmark of mh_gemm on the shapes of BASELINE config 2 (B=16): TFLOP/s per shape, HIP-event timed."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirror_amd import kernels as K
from mirror_amd._lib import MH_BF16, MH_F32, ACT_RELU

dev = "cuda"
bf, f32 = torch.bfloat16, torch.float32
B, n_p, D, h, dh, m, N, F = 16, 4352, 512, 8, 64, 256, 4096, 1024


def rn(*s, dt=bf):
    return (torch.randn(*s, device=dev) * 0.5).to(dt)


def heads(t, which, parts):
    Bn, T, Dt = t.shape
    d = Dt // parts // h
    return t.view(Bn, T, parts, h, d)[:, :, which].permute(0, 2, 1, 3)


def timeit(name, fn, flops, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:44s} {ms*1e3:9.1f} us  {flops/ms/1e9:8.1f} TF/s")


x = rn(B * n_p, D); wqkv = rn(3 * D, D); qkv = rn(B, n_p, 3 * D); lm = rn(B, m, 2 * D)
timeit("to_qkv fwd  NT  [69632x512]x[512x1536] bf16", lambda: K.gemm(x, wqkv.t(), mma=MH_BF16), 2 * B * n_p * D * 3 * D)
dy = rn(B * n_p, 3 * D)
timeit("to_qkv dX   NN  [69632x1536]x[1536x512]", lambda: K.gemm(dy, wqkv, mma=MH_BF16), 2 * B * n_p * D * 3 * D)
dw = torch.zeros(3 * D, D, device=dev)
timeit("to_qkv dW   TN  split-K 16 -> f32 atomics", lambda: K.gemm(dy.t(), x, out=dw, accumulate=True, split_k=21, mma=MH_BF16), 2 * B * n_p * D * 3 * D)
wsi = rn(B, N, F); w1 = rn(D, F); b1 = torch.zeros(D, device=dev); seq = torch.empty(B, N + 65, D, device=dev)
timeit("_fc1 fwd    NT  16x[4096x1024]x[1024x512]+relu f32", lambda: K.gemm(wsi, w1.t(), out=seq[:, 1:1 + N], bias=b1, act=ACT_RELU, mma=MH_BF16), 2 * B * N * F * D)
q, k, v = (heads(qkv, i, 3) for i in range(3)); ql, kl = heads(lm, 0, 2), heads(lm, 1, 2)
timeit("sim1  q.kl^T   128x[4352x64]x[64x256] -> f32", lambda: K.gemm(q, kl.transpose(-1, -2), alpha=.125, mma=MH_BF16, out_dtype=f32), 2 * B * h * n_p * dh * m)
timeit("sim3  ql.k^T   128x[256x64]x[64x4352] -> f32", lambda: K.gemm(ql, k.transpose(-1, -2), alpha=.125, mma=MH_BF16, out_dtype=f32), 2 * B * h * n_p * dh * m)
a1 = rn(B, h, n_p, m); a3 = rn(B, h, m, n_p); w2 = rn(B, h, m, dh); out = torch.empty(B, n_p, D, device=dev, dtype=bf)
timeit("a3.v          128x[256x4352]x[4352x64] -> f32", lambda: K.gemm(a3, v, mma=MH_BF16, out_dtype=f32), 2 * B * h * n_p * dh * m)
timeit("a1.w2         128x[4352x256]x[256x64] -> bf16 cols", lambda: K.gemm(a1, w2, out=heads(out, 0, 1), mma=MH_BF16), 2 * B * h * n_p * dh * m)
dO = heads(rn(B, n_p, D), 0, 1)
timeit("dS1 = dO.w2^T 128x[4352x64]x[64x256] -> bf16", lambda: K.gemm(dO, w2.transpose(-1, -2), mma=MH_BF16), 2 * B * h * n_p * dh * m)
timeit("dW2 = a1^T.dO 128x[256x4352]x[4352x64] TN", lambda: K.gemm(a1.transpose(-1, -2), dO, mma=MH_BF16, out_dtype=f32), 2 * B * h * n_p * dh * m)
timeit("dv  = a3^T.dAV 128x[4352x256]x[256x64] TN", lambda: K.gemm(a3.transpose(-1, -2), w2, out=heads(torch.empty_like(qkv), 2, 3), mma=MH_BF16), 2 * B * h * n_p * dh * m)
a2 = rn(B, h, m, m, dt=f32); z = rn(B, h, m, m, dt=f32)
timeit("pinv a2.z     128x[256^3] f32 MFMA", lambda: K.gemm(a2, z, mma=MH_F32), 2 * B * h * m ** 3)
timeit("pinv a2.z     128x[256^3] f32 in, bf16 MFMA", lambda: K.gemm(a2, z, mma=MH_BF16), 2 * B * h * m ** 3)
timeit("pinv z^T.dz   128x[256^3] TN f32 in, bf16 MFMA", lambda: K.gemm(z.transpose(-1, -2), a2, mma=MH_BF16), 2 * B * h * m ** 3)
timeit("pinv dz.T3^T  128x[256^3] NT f32 in, bf16 MFMA", lambda: K.gemm(a2, z.transpose(-1, -2), mma=MH_BF16), 2 * B * h * m ** 3)
xr = rn(16, 2048); wr = rn(1024, 2048)
timeit("rna fc1       [16x2048]x[2048x1024]", lambda: K.gemm(xr, wr.t(), mma=MH_BF16), 2 * 16 * 2048 * 1024)
