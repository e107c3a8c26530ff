#!/usr/bin/env python3
"""mh_layernorm_fwd_lm alone at the c4 / c2 layer shapes, 1 / 2 / 4 waves per landmark group (MH_LN_LM_SPLIT): us per launch and
equality of the outputs with the one-wave form (rows bit-equal; landmark rows to f32 summation order)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mirror_amd import kernels as K  # noqa: E402
dev = torch.device("cuda", 0)
for name, (B, rows, D, l, masked) in {"c4": (8, 8282, 512, 33, True), "c2": (16, 4097, 512, 17, False), "c4_B16": (16, 8282, 512, 33, True)}.items():
    m = -(-rows // l); m = 256
    pad = m * l - rows
    torch.manual_seed(0)
    x = torch.randn(B, rows, D, device=dev)
    g, b = torch.randn(D, device=dev), torch.randn(D, device=dev)
    rm = ls = None
    if masked:
        mask = torch.rand(B, rows, device=dev) > 0.2
        rm, _, ls = K.keymask_plan(mask.contiguous(), 0, 0, pad, l)
    ref = None
    for split in ("1", "2", "4"):
        os.environ["MH_LN_LM_SPLIT"] = split
        y = torch.empty(B, pad + rows, D, device=dev, dtype=torch.bfloat16)
        mean, rstd = torch.empty(B * rows, device=dev), torch.empty(B * rows, device=dev)
        xpm = torch.empty(B, m, D, device=dev)
        xpm16 = torch.empty(B, m, D, device=dev, dtype=torch.bfloat16)
        def run():
            K.layernorm_fwd_lm(x, g, b, y, mean, rstd, xpm, B, rows, D, rows * D, pad, l, 1e-5, xpm_bf16=xpm16, row_mask=rm, lm_scale=ls)
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record()
            for _ in range(20):
                run()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        byts = x.numel() * 4 + y.numel() * 2
        out = (y.clone(), mean.clone(), rstd.clone(), xpm.clone())
        if ref is None:
            ref = out
        eq = [torch.equal(a, c) for a, c in zip(out[:3], ref[:3])]
        dl = float((out[3] - ref[3]).abs().max() / ref[3].abs().max().clamp_min(1e-9)) if not masked else float(((out[3] - ref[3]).abs() / (ref[3].abs() + 1e-3)).max())
        print(f"{name} split {split}: {min(ts):7.1f} us  ({byts / min(ts) / 1e6:.2f} TB/s)  rows equal {eq}  landmark rel diff {dl:.2e}", flush=True)
os.environ.pop("MH_LN_LM_SPLIT", None)
