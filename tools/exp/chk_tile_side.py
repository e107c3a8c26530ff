#!/usr/bin/env python3
"""TransLayer(768) forward + backward under the four settings of Fn._TILE_SIDE / Fn._PINV_R32, twice each: relative differences of dx."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mirror_amd import functional as Fn
MM = importlib.import_module("mirror_amd.models.mirror")
prec = Fn.POLICIES["bf16"]
g = torch.Generator().manual_seed(21)
x0 = torch.randn(2, 900, 768, generator=g).cuda()
up = torch.randn(2, 900, 768, generator=g).cuda()


def run(side, r32):
    Fn._TILE_SIDE, Fn._PINV_R32 = side, r32
    torch.manual_seed(3)
    layer = MM.TransLayer(768).cuda().eval()
    x = x0.clone().requires_grad_(True)
    y = layer(x, prec)
    y.backward(up.clone())      # (the residual add hands its upstream gradient on in place)
    torch.cuda.synchronize()
    return y.detach().float(), x.grad.float()


ref = run(False, False)
for side, r32 in ((False, False), (False, True), (True, False), (True, True), (True, True)):
    y, dx = run(side, r32)
    print(f"side={side} r32={r32}: y equal {torch.equal(y, ref[0])}, dx rel diff {float((dx - ref[1]).norm() / ref[1].norm()):.3e}", flush=True)
