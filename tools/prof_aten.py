#!/usr/bin/env python3
"""aten ops (torch glue) of one eager training step with their input shapes: which tensors does autograd still sum / copy / fill?"""
import os, sys, collections
os.environ["MIRROR_GRAPH"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import mirror_amd.models as M
from mirror_amd.engine import TrainEngine
from mirror_amd.losses import MIRRORLoss
dev = torch.device("cuda", 0)
torch.manual_seed(42)
model = M.mirror(wsi_embed_dim=1024, rna_embed_dim=2048, embed_dim=512, wsi_num_tokens=4096, rna_encoder_depth=6, rna_mlp_ratio=4.0,
                 rna_norm_layer="layernorm", rna_act_layer="gelu", rna_num_heads=8).to(dev).train()
eng = TrainEngine(model, MIRRORLoss(), lr=2e-5, precision="bf16")
wsi = torch.randn(16, 4096, 1024, device=dev).bfloat16()
rna = torch.randn(16, 2048, device=dev)
for _ in range(3):
    eng.step(wsi, rna)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    eng.step(wsi, rna)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.name.split("::")[1] in ("add", "add_", "mul", "copy_", "fill_", "zero_", "cat", "contiguous", "clone", "sum", "div", "neg", "to", "_to_copy"):
        cnt[(e.name, str(e.input_shapes)[:90])] += 1
for (n, sh), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:50]:
    print(f"{c:4d} {n:18s} {sh}")
