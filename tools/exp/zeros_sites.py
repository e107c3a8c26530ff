#!/usr/bin/env python3
"""Which call sites carve how much of the step's zero arena (Fn.zeros) in one c2 step."""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mirror_amd.models as M
from mirror_amd.engine import TrainEngine
from mirror_amd.losses import MIRRORLoss
from mirror_amd import functional as Fn
dev = torch.device("cuda", 0)
torch.manual_seed(42)
model = M.mirror(wsi_embed_dim=1024, rna_embed_dim=2048, embed_dim=512, wsi_num_tokens=4096, rna_encoder_depth=6, rna_mlp_ratio=4.0,
                 rna_norm_layer="layernorm", rna_act_layer="gelu", rna_num_heads=8).to(dev).train()
eng = TrainEngine(model, MIRRORLoss(alignment_loss_weight=0.5, wsi_retention_loss_weight=0.15, rna_retention_loss_weight=0.15,
                                    style_loss_weight=0.1, cluster_loss_weight=0.1), lr=2e-5, precision="bf16", graph=False)
g = torch.Generator(device=dev).manual_seed(1234)
wsi = torch.randn(16, 4096, 1024, device=dev, generator=g).to(torch.bfloat16)
rna = torch.randn(16, 2048, device=dev, generator=g)
eng.step(wsi, rna); eng.step(wsi, rna)
sites = collections.Counter()
real = Fn.zeros


def logged(shape, device):
    shape_t = (shape,) if isinstance(shape, int) else tuple(shape)
    n = 1
    for d in shape_t:
        n *= int(d)
    fr = traceback.extract_stack(limit=2)[0]
    sites[(os.path.basename(fr.filename), fr.lineno, fr.name)] += n * 4
    return real(shape, device)


Fn.zeros = logged
eng.step(wsi, rna)
torch.cuda.synchronize()
tot = sum(sites.values())
print(f"total {tot / 1e6:.1f} MB over {len(sites)} sites")
for (f, ln, fn), b in sites.most_common(14):
    print(f"{b / 1e6:9.2f} MB  {f}:{ln} {fn}")
