#!/usr/bin/env python3
"""The step's fused-epilogue / f32-output projections at c2 shapes, each alone on the chip, against the plain bf16-output product of
the same shape (what the epilogue costs on top of the K loop).  usage: PYTHONPATH=. python tools/exp/time_epi_gemms.py"""
import torch
from mirror_amd import kernels as K
from mirror_amd._lib import ACT_NONE, ACT_RELU, MH_BF16
dev, bf, f32 = "cuda", torch.bfloat16, torch.float32
B, T, D, F = 16, 4097, 512, 1024
g = torch.Generator(device=dev).manual_seed(0)
M = B * T
Mt = (M // 256) * 256


def t(name, fn, flops, nbytes, reps=20):
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
    print(f"{name:52s} {ms * 1e3:7.1f} us  {flops / ms / 1e9:7.1f} TF/s  {nbytes / ms / 1e9:6.2f} TB/s", flush=True)


x = (torch.randn(B, T, D, device=dev, generator=g) * 0.5).to(bf)
w = (torch.randn(D, D, device=dev, generator=g) * 0.05).to(bf)
bias = torch.randn(D, device=dev, generator=g)
fl = 2.0 * Mt * D * D
ob = torch.empty(Mt, D, device=dev, dtype=bf)
of = torch.empty(M, D, device=dev, dtype=f32)
of2 = of[:Mt]
x2 = x.view(M, D)[:Mt]
t("plain  [65536 x 512] x [512 x 512] -> bf16", lambda: K.gemm(x2, w.t(), out=ob, mma=MH_BF16), fl, Mt * D * 4)
t("plain  ... + bias -> bf16", lambda: K.gemm(x2, w.t(), out=ob, bias=bias, mma=MH_BF16), fl, Mt * D * 4)
t("plain  ... -> f32", lambda: K.gemm(x2, w.t(), out=of2, mma=MH_BF16), fl, Mt * D * 6)
t("plain  ... + bias -> f32", lambda: K.gemm(x2, w.t(), out=of2, bias=bias, mma=MH_BF16), fl, Mt * D * 6)
resid = torch.randn(M, D, device=dev, generator=g)
t("to_out: + bias, dropout, + resid -> f32 (DROPADD)", lambda: K.linear_fused(x, w, bias, of, K.epi_dropadd(resid, 0.1, 1234, 0, None), m_rows=Mt), fl, Mt * D * 10)
mask = (torch.rand(B, T - 1, device=dev, generator=g) < 0.75).float()
tok = torch.randn(D, device=dev, generator=g)
pos = torch.randn(T * D, device=dev, generator=g)
t("retention_embed: + bias, mask token, + pos -> f32 (MASKPOS)", lambda: K.linear_fused(x, w, bias, of, K.epi_maskpos(mask, tok, pos, T, 1), m_rows=Mt), fl, Mt * D * 6)
R = T - 1
xw = x
tgt = torch.randn(B, T, D, device=dev, generator=g)
acc = torch.zeros(2, device=dev)
oh = torch.empty(B * R, D, device=dev, dtype=bf)
t("retention_head rows 1..: + bias, sq. error vs f32 target -> bf16 (SQERR)", lambda: K.linear_fused(xw, w, bias, oh, K.epi_sqerr(mask, tgt[:, 1:], tgt.stride(0), acc, R), window=(1, R)), 2.0 * B * R * D * D, B * R * D * 8)
wsi = (torch.randn(B, 4096, F, device=dev, generator=g)).to(bf)
w1 = (torch.randn(D, F, device=dev, generator=g) * 0.03).to(bf)
seq = torch.empty(B, 4097, D, device=dev, dtype=f32)
t("_fc1: [65536 x 1024] x [1024 x 512] + bias, ReLU -> f32 rows 1..", lambda: K.gemm(wsi, w1.t(), out=seq[:, 1:], bias=bias, act=ACT_RELU, mma=MH_BF16), 2.0 * B * 4096 * F * D, B * 4096 * (F * 2 + D * 4))
w3 = (torch.randn(2 * D, D, device=dev, generator=g) * 0.05).to(bf)
o3 = torch.empty(Mt, 2 * D, device=dev, dtype=bf)
t("q|k: [65536 x 512] x [512 x 1024] -> bf16", lambda: K.gemm(x2, w3.t(), out=o3, mma=MH_BF16), 2 * fl, Mt * D * 2 * 3)
