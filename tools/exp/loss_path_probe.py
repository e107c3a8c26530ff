#!/usr/bin/env python3
"""Which MIRRORLoss form does the benched c2 / bf16 step take, and what dtypes reach the loss?"""
import torch
import mirror_amd.models as M
from mirror_amd import functional as Fn
from mirror_amd.losses import MIRRORLoss
from oracle.mirror_oracle import OUTPUT_NAMES

dev = "cuda"
torch.manual_seed(42)
model = M.mirror(wsi_embed_dim=1024, rna_embed_dim=2048, embed_dim=512, wsi_num_tokens=4096, rna_encoder_depth=2, rna_mlp_ratio=2.572,
                 rna_norm_layer="layernorm", rna_act_layer="gelu", rna_num_heads=8).to(dev).train()
model.precision = "bf16"
outs = model(torch.randn(2, 4096, 1024, device=dev).bfloat16(), torch.randn(2, 2048, device=dev))
for n, o in zip(OUTPUT_NAMES, outs):
    print(f"{n:24s} {str(o.dtype):16s} {tuple(o.shape)} contiguous={o.is_contiguous()} grad={o.requires_grad}")
out = MIRRORLoss()(*outs)
print(out[0].grad_fn)
