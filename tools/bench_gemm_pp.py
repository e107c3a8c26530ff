#!/usr/bin/env python3
"""A/B of the two 256 x 256-tile GEMM main loops (mh_gemm_select_pp) on the step's large shapes, interleaved in one process:
correctness against an f32 torch product on a row sample, then microseconds / TFLOP/s per shape and kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirror_amd import _lib, kernels as K          # noqa: E402
from mirror_amd._lib import ACT_RELU, MH_BF16      # noqa: E402

lib = _lib.load()
dev, bf, f32 = "cuda", torch.bfloat16, torch.float32


def rn(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(bf)


def check(name, got, a, b, rows=64):
    idx = torch.randint(0, a.shape[0], (rows,), device=dev)
    ref = a[idx].float() @ b.float()
    err = float((got[idx].float() - ref).abs().max()) / max(float(ref.abs().max()), 1e-9)
    return err


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


MODES = ((2, "pq"), (1, "pp"), (0, "reg"))


def run(name, make, flops, ref=None):
    res = {}
    for rnd in range(2):
        for mode, _ in MODES:
            lib.mh_gemm_select_pp(mode)
            fn = make()
            out = fn()
            torch.cuda.synchronize()
            err = ref(out) if (ref is not None and rnd == 0) else None
            us = timeit(fn)
            res.setdefault(mode, []).append((us, err))
    lib.mh_gemm_select_pp(2)
    best = {m: min(u for u, _ in res[m]) for m, _ in MODES}
    line = f"{name:58s}"
    for m, nm in MODES:
        line += f" {nm} {best[m]:7.1f} us {flops / best[m] / 1e6:5.0f} TF/s |"
    line += f" pq/reg x{best[0] / best[2]:4.2f} | err " + " ".join(f"{res[m][0][1]:.1e}" for m, _ in MODES)
    print(line, flush=True)


M, D = 69632, 512
x = rn(M, D); wqkv = rn(3 * D, D); dy = rn(M, 3 * D)
run("to_qkv q|k fwd  KC,KC [69632x512]x[512x1024] bf16", lambda: (lambda: K.gemm(x, wqkv[:1024].t(), mma=MH_BF16)), 2 * M * D * 1024,
    lambda o: check("", o, x, wqkv[:1024].t()))
run("to_qkv v   fwd  KC,KC [69632x512]x[512x512] bf16", lambda: (lambda: K.gemm(x, wqkv[1024:].t(), mma=MH_BF16)), 2 * M * D * 512,
    lambda o: check("", o, x, wqkv[1024:].t()))
run("to_qkv dgrad    KC,KS [69632x1536]x[1536x512] bf16", lambda: (lambda: K.gemm(dy, wqkv, mma=MH_BF16)), 2 * M * D * 1536,
    lambda o: check("", o, dy, wqkv))
dw = torch.zeros(3 * D, D, device=dev)


def wgrad():
    dw.zero_()
    return K.gemm(dy.t(), x, out=dw, accumulate=True, split_k=21, mma=MH_BF16)


run("to_qkv wgrad    KS,KS [1536x69632]x[69632x512] split 21", lambda: wgrad, 2 * M * D * 1536,
    lambda o: float((o - dy.float().t() @ x.float()).abs().max()) / float((dy.float().t() @ x.float()).abs().max()))
M2 = 65536
xo = rn(M2, D); wo = rn(D, D); bo = torch.randn(D, device=dev)
run("to_out fwd      KC,KC [65536x512]x[512x512] bf16 + bias", lambda: (lambda: K.gemm(xo, wo.t(), bias=bo, mma=MH_BF16)), 2 * M2 * D * D,
    lambda o: float((o[:64].float() - (xo[:64].float() @ wo.float().t() + bo)).abs().max()))
run("to_out dgrad    KC,KS [65536x512]x[512x512] bf16", lambda: (lambda: K.gemm(xo, wo, mma=MH_BF16)), 2 * M2 * D * D, lambda o: check("", o, xo, wo))
dwo = torch.zeros(D, D, device=dev)


def wgrad2():
    dwo.zero_()
    return K.gemm(xo.t(), xo, out=dwo, accumulate=True, split_k=64, mma=MH_BF16)


run("to_out wgrad    KS,KS [512x65536]x[65536x512] split 64", lambda: wgrad2, 2 * M2 * D * D,
    lambda o: float((o - xo.float().t() @ xo.float()).abs().max()) / float((xo.float().t() @ xo.float()).abs().max()))
wsi = rn(16, 4096, 1024); w1 = rn(D, 1024); b1 = torch.zeros(D, device=dev); seq = torch.empty(16, 4096 + 65, D, device=dev)
run("_fc1 fwd        KC,KC 16x[4096x1024]x[1024x512] f32 + relu", lambda: (lambda: K.gemm(wsi, w1.t(), out=seq[:, 1:4097], bias=b1, act=ACT_RELU, mma=MH_BF16)),
    2 * 65536 * 1024 * D, lambda o: float((o[0, :64] - torch.relu(wsi[0, :64].float() @ w1.float().t())).abs().max()))
dh = rn(65536, D); xf = wsi.reshape(65536, 1024); dw1 = torch.zeros(D, 1024, device=dev)


def wgrad3():
    dw1.zero_()
    return K.gemm(dh.t(), xf, out=dw1, accumulate=True, split_k=32, mma=MH_BF16)


run("_fc1 wgrad      KS,KS [512x65536]x[65536x1024] split 32", lambda: wgrad3, 2 * 65536 * 1024 * D,
    lambda o: float((o - dh.float().t() @ xf.float()).abs().max()) / float((dh.float().t() @ xf.float()).abs().max()))
for n in (4096, 8192):
    a, b = rn(n, n), rn(n, n)
    run(f"square          KC,KC {n}^3 bf16", lambda: (lambda: K.gemm(a, b.t(), mma=MH_BF16)), 2 * n ** 3, lambda o: check("", o, a, b.t()))
    run(f"square          KC,KS {n}^3 bf16", lambda: (lambda: K.gemm(a, b, mma=MH_BF16)), 2 * n ** 3, lambda o: check("", o, a, b))
