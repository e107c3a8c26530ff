#!/usr/bin/env python3
"""Which of round 5's hand-over fusions does one eager c2 step actually take (spies on the kernel wrappers)?"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mirror_amd.models as M
from mirror_amd.engine import TrainEngine
from mirror_amd.losses import MIRRORLoss
from mirror_amd import kernels as K
seen = collections.Counter()
real_ln, real_mse, real_ma, real_dc, real_adam = K.layernorm_bwd, K.mse_masked_bwd, K.mask_apply_bwd, K.dropout_lite_colsum, K.adam


def ln(*a, **kw):
    for k in ("fan", "drop", "relu_out", "relu_db"):
        if kw.get(k) is not None:
            seen["layernorm_bwd " + k] += 1
    return real_ln(*a, **kw)


def mse(*a, **kw):
    seen["mse_masked_bwd colsum_ws" if kw.get("colsum_ws") is not None else "mse_masked_bwd plain"] += 1
    return real_mse(*a, **kw)


def ma(*a, **kw):
    seen["mask_apply_bwd dbias" if kw.get("dbias") is not None else "mask_apply_bwd plain"] += 1
    return real_ma(*a, **kw)


def dc(*a, **kw):
    seen["dropout_lite_colsum (standalone)"] += 1
    return real_dc(*a, **kw)


def adam(*a, **kw):
    seen[f"adam tick={kw.get('tick', True)} hole={'yes' if kw.get('hole') else 'no'}"] += 1
    return real_adam(*a, **kw)


K.layernorm_bwd, K.mse_masked_bwd, K.mask_apply_bwd, K.dropout_lite_colsum, K.adam = ln, mse, ma, dc, adam
dev = torch.device("cuda", 0)
torch.manual_seed(42)
model = M.mirror(wsi_embed_dim=1024, rna_embed_dim=2048, embed_dim=512, wsi_num_tokens=4096, rna_encoder_depth=6, rna_mlp_ratio=4.0,
                 rna_norm_layer="layernorm", rna_act_layer="gelu", rna_num_heads=8).to(dev).train()
eng = TrainEngine(model, MIRRORLoss(), lr=2e-5, precision="bf16", graph=False)
g = torch.Generator(device=dev).manual_seed(1234)
wsi = torch.randn(16, 4096, 1024, device=dev, generator=g).to(torch.bfloat16)
rna = torch.randn(16, 2048, device=dev, generator=g)
if os.environ.get("MIRROR_PROBE"):
    print("(with probes)")
eng.step(wsi, rna)
torch.cuda.synchronize()
for k, v in sorted(seen.items()):
    print(f"{v:3d} x {k}")
