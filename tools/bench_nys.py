#!/usr/bin/env python3
"""Fused Nystrom attention sides in isolation at the c2 geometry (B=16, h=8, n_p=4352, m=256, dh=64)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirror_amd import kernels as K
dev, bf = "cuda", torch.bfloat16
B, h, n_p, m, dh = 16, 8, 4352, 256, 64
D = h * dh
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(B, n_p, 3 * D, device=dev, generator=g) * 0.5).to(bf)
lm = (torch.randn(B, m, 2 * D, device=dev, generator=g) * 0.5).to(bf)
w2 = (torch.randn(B, h, m, dh, device=dev, generator=g) * 0.1).to(bf)
out = torch.zeros(B, n_p, D, device=dev, dtype=bf)
dout = (torch.randn(B, n_p, D, device=dev, generator=g) * 0.1).to(bf)
dav = (torch.randn(B, h, m, dh, device=dev, generator=g) * 0.1).to(bf)
dqkv = torch.empty_like(qkv)
dw2 = torch.zeros(B, h, m, dh, device=dev)
dlm = torch.zeros(B, m, 2 * D, device=dev)
scale = dh ** -0.5
o1 = torch.empty_like(out)
lse1 = K.nys_attn1_fwd(qkv, lm, w2, out, h, scale, o1=o1)
delta1 = torch.empty_like(lse1)
av, lse3 = K.nys_attn3_fwd(qkv, lm, h, scale)
unit = 2.0 * n_p * m * dh * B * h / 1e9     # one [n_p x m x dh] product over every (b, h), GFLOP


def t(name, fn, products, reps=10):
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
    print(f"{name:28s} {ms * 1e3:8.1f} us   {products * unit / ms:7.1f} TF/s", flush=True)


t("attn1 fwd", lambda: K.nys_attn1_fwd(qkv, lm, w2, out, h, scale), 2)
t("attn1 fwd (accumulate)", lambda: K.nys_attn1_fwd(qkv, lm, w2, out, h, scale, accumulate=True), 2)
t("attn3 fwd", lambda: K.nys_attn3_fwd(qkv, lm, h, scale), 2)
t("attn1 fwd (accumulate, + o1)", lambda: K.nys_attn1_fwd(qkv, lm, w2, out, h, scale, accumulate=True, o1=o1), 2)
t("attn1 bwd part 1 (dw2, dk_l, delta)", lambda: K.nys_attn1_bwd(qkv, lm, w2, dout, lse1, o1, delta1, dqkv, dw2, dlm, h, scale, which=1), 4)
t("attn1 bwd part 2 (dq)", lambda: K.nys_attn1_bwd(qkv, lm, w2, dout, lse1, o1, delta1, dqkv, dw2, dlm, h, scale, which=2), 3)
t("attn3 bwd (delta+dkv+dql)", lambda: K.nys_attn3_bwd(qkv, lm, av, dav, lse3, dqkv, dlm, h, scale, one_pass=False), 7)
t("attn3 bwd (delta + ONE pass)", lambda: K.nys_attn3_bwd(qkv, lm, av, dav, lse3, dqkv, dlm, h, scale, one_pass=True), 5)

# res_conv: alone, inside attn3's forward, and its two gradients as one pass (round 5)
w = (torch.randn(h, 1, 33, 1, device=dev, generator=g) * 0.2)
wflat = w.reshape(-1).contiguous()
dres = torch.zeros(h * 33, device=dev)
t("res_conv fwd (own launch)", lambda: K.resconv(qkv[..., 2 * D:], w, out, h, transpose=False, accumulate=False), 0)
t("attn3 fwd + res_conv inside", lambda: K.nys_attn3_fwd(qkv, lm, h, scale, rc=(wflat, out)), 2)
t("res_conv adjoint (own launch)", lambda: K.resconv(dout, w, dqkv[..., 2 * D:], h, transpose=True, accumulate=True), 0)
t("res_conv tap gradient (own)", lambda: K.resconv_wgrad(qkv[..., 2 * D:], dout, dres, h), 0)
t("res_conv both gradients, 1 pass", lambda: K.resconv_bwd(dout, qkv[..., 2 * D:], w, dqkv[..., 2 * D:], dres, h), 0)
