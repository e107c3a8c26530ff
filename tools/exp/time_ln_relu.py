#!/usr/bin/env python3
"""Isolated timing of the c2-sized backward passes round 5 fused bias gradients / the fan-out sum into (run from a tree's root)."""
import os, sys, inspect
sys.path.insert(0, os.getcwd())
import torch
from mirror_amd import kernels as K
dev = "cuda"
B, T, D, l = 16, 4097, 512, 16
pad = (l - T % l) % l
m = (pad + T) // l
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, T, D, device=dev, generator=g); x[:, 1:].clamp_(min=0)
gam = torch.randn(D, device=dev, generator=g)
mean, rstd = x.mean(-1).reshape(-1).contiguous(), (x.var(-1, unbiased=False) + 1e-5).rsqrt().reshape(-1).contiguous()
dy = torch.randn(B, pad + T, D, device=dev, generator=g).to(torch.bfloat16)
gadd = torch.randn(B, m, D, device=dev, generator=g).to(torch.bfloat16)
G = torch.zeros(B, T, D, device=dev); dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
dh = torch.empty(B, T - 1, D, device=dev, dtype=torch.bfloat16); rdb = torch.zeros(D, device=dev)
has_db = "relu_db" in inspect.signature(K.layernorm_bwd).parameters


def timeit(f, n=40):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def ln_relu(with_db):
    kw = dict(relu_db=rdb) if (with_db and has_db) else {}
    K.layernorm_bwd(dy[:, pad:], x, gam, mean, rstd, G, dg, db, B, T, D, T * D, (pad + T) * D, accumulate_dx=True, gadd=gadd, pad=pad, l=l,
                    relu_out=dh, relu_first=1, **kw)


print(f"LN bwd lm + relu_out: {timeit(lambda: ln_relu(False)):.1f} us" + (f"; + relu_db: {timeit(lambda: ln_relu(True)):.1f} us; again without {timeit(lambda: ln_relu(False)):.1f}, with {timeit(lambda: ln_relu(True)):.1f}" if has_db else ""))
print(f"colsum bf16 [B*N, D]: {timeit(lambda: K.colsum(dh.reshape(-1, D), rdb)):.1f} us")
# final norm: fan-out
gf = torch.randn(B, T, D, device=dev, generator=g); src = torch.randn(B, T - 1, D, device=dev, generator=g).to(torch.bfloat16)
cls = torch.randn(B, D, device=dev, generator=g); dx = torch.empty_like(x)
def plain():
    dE = K.fanout_bwd(gf, src, -1.0, cls, B, T, D)
    K.layernorm_bwd(dE, x, gam, mean, rstd, dx, dg, db, B, T, D, T * D, T * D)
print(f"fanout + LN bwd: {timeit(plain):.1f} us", end="")
if "fan" in inspect.signature(K.layernorm_bwd).parameters:
    print(f"; LN bwd with the fan-out inside: {timeit(lambda: K.layernorm_bwd(gf, x, gam, mean, rstd, dx, dg, db, B, T, D, T * D, T * D, fan=(src, -1.0, cls))):.1f} us")
else:
    print()
# dropout backward + bias gradient inside the LayerNorm backward
if "drop" in inspect.signature(K.layernorm_bwd).parameters:
    gbo = torch.empty(B, T, D, device=dev, dtype=torch.bfloat16); dbo = torch.zeros(D, device=dev)
    def two():
        K.layernorm_bwd(gf, x, gam, mean, rstd, dx, dg, db, B, T, D, T * D, T * D, fan=(src, -1.0, cls))
        K.dropout_lite_colsum(dx, 0.1, 77, 64, None, gbo, dbo)
    def one():
        K.layernorm_bwd(gf, x, gam, mean, rstd, dx, dg, db, B, T, D, T * D, T * D, fan=(src, -1.0, cls), drop=(gbo, 0.1, 77, 64, None, dbo))
    print(f"LN bwd (fan) + dropout_lite_colsum: {timeit(two):.1f} us; one launch: {timeit(one):.1f} us")
    dyb = dy[:, pad:pad + T].contiguous()
    def two_b():
        K.layernorm_bwd(dyb, x, gam, mean, rstd, dx, dg, db, B, T, D, T * D, T * D)
        K.dropout_lite_colsum(dx, 0.1, 77, 64, None, gbo, dbo)
    def one_b():
        K.layernorm_bwd(dyb, x, gam, mean, rstd, dx, dg, db, B, T, D, T * D, T * D, drop=(gbo, 0.1, 77, 64, None, dbo))
    print(f"LN bwd (bf16 dy) + dropout_lite_colsum: {timeit(two_b):.1f} us; one launch: {timeit(one_b):.1f} us")
# masked MSE backward
pred = torch.randn(B, T - 1, D, device=dev, generator=g).to(torch.bfloat16); E = torch.randn(B, T, D, device=dev, generator=g)
mask = (torch.rand(B, T - 1, device=dev, generator=g) < 0.75).float(); acc = torch.zeros(2, device=dev)
K.mse_masked_fwd(pred, E[:, 1:], mask, acc, B * (T - 1), D)
dp = torch.empty_like(pred); one = torch.ones(1, device=dev)
def mse_plain():
    K.mse_masked_bwd(pred, E[:, 1:], mask, acc, one, dp, None, B * (T - 1), D)
    K.colsum(dp.reshape(-1, D), rdb)
print(f"mse bwd + colsum: {timeit(mse_plain):.1f} us", end="")
if hasattr(K, "MSE_CS_BLOCKS"):
    ws = torch.empty(K.MSE_CS_BLOCKS, D, device=dev)
    def mse_cs():
        K.mse_masked_bwd(pred, E[:, 1:], mask, acc, one, dp, None, B * (T - 1), D, colsum_ws=ws)
        K.colsum(ws, rdb)
    print(f"; mse bwd with column sums + fold: {timeit(mse_cs):.1f} us")
else:
    print()
# mask/pos backward
dyf = torch.randn(B, T, D, device=dev, generator=g); dr = torch.empty(B, T, D, device=dev, dtype=torch.bfloat16)
dtok, dpos = torch.zeros(D, device=dev), torch.zeros(T * D, device=dev)
def ma_plain():
    K.mask_apply_bwd(dyf, mask, dtok, dpos, B, T, D, 1, False, out=dr)
    K.colsum(dr.reshape(-1, D), rdb)
print(f"mask/pos bwd + colsum: {timeit(ma_plain):.1f} us", end="")
if "dbias" in inspect.signature(K.mask_apply_bwd).parameters:
    print(f"; with dbias: {timeit(lambda: K.mask_apply_bwd(dyf, mask, dtok, dpos, B, T, D, 1, False, out=dr, dbias=rdb)):.1f} us")
else:
    print()
