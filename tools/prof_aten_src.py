#!/usr/bin/env python3
"""aten ops (torch glue) of one eager training step WITH the mirror_amd source line that issued them (torch profiler, with_stack)."""
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
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    eng.step(wsi, rna)
    torch.cuda.synchronize()
LAUNCHING = ("add", "add_", "mul", "mul_", "copy_", "fill_", "zero_", "cat", "sum", "div", "neg", "exp", "clamp_", "rand", "randn", "uniform_", "normal_", "sub", "where", "index", "masked_fill_")
cnt = collections.Counter()
for e in prof.events():
    if not e.name.startswith("aten::") or e.name.split("::")[1] not in LAUNCHING:
        continue
    src = "(autograd engine / no python frame)"
    for fr in (e.stack or []):
        if "mirror_amd" in fr and "site-packages" not in fr:
            src = fr.strip().split("/root/repo/")[-1] if "/root/repo/" in fr else fr.strip()
            break
    cnt[(e.name, str(e.input_shapes)[:60], src[:110])] += 1
for (n, sh, src), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{c:3d} {n:14s} {sh:60s} {src}")
