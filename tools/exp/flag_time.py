#!/usr/bin/env python3
"""ms per replayed c2 step with module-level test hooks flipped, one process per setting (the step is captured once per process).
usage: python3 tools/exp/flag_time.py [steps] [module.NAME=value ...]   e.g. functional._TAIL_ASIDE=False engine._DEFER_SKINNY=False"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
args = sys.argv[1:]
steps = int(args.pop(0)) if args and args[0].isdigit() else 30
import mirror_amd.models as M                     # noqa: E402
from mirror_amd.engine import TrainEngine         # noqa: E402
from mirror_amd.losses import MIRRORLoss          # noqa: E402
from mirror_amd import functional as Fn           # noqa: E402

for a in args:
    name, val = a.split("=")
    mod, attr = name.rsplit(".", 1)
    m = importlib.import_module("mirror_amd." + mod)
    assert hasattr(m, attr), name
    setattr(m, attr, eval(val))
dev = torch.device("cuda", 0)
torch.manual_seed(42)
c4 = os.environ.get("FLAG_CONFIG", "c2") == "c4"       # BASELINE config 4: 8192 x 768-d tokens, key-padding mask, B = 8
N, F, B = (8192, 768, 8) if c4 else (4096, 1024, 16)
model = M.mirror(wsi_embed_dim=F, rna_embed_dim=2048, embed_dim=512, wsi_num_tokens=N, rna_encoder_depth=6, rna_mlp_ratio=4.0,
                 rna_norm_layer="layernorm", rna_act_layer="gelu", rna_num_heads=8).to(dev).train()
eng = TrainEngine(model, MIRRORLoss(alignment_loss_weight=0.5, wsi_retention_loss_weight=0.15, rna_retention_loss_weight=0.15,
                                    style_loss_weight=0.1, cluster_loss_weight=0.1), lr=2e-5, precision="bf16")
Fn.manual_seed(1234)
g = torch.Generator(device=dev).manual_seed(1234)
wsi = torch.randn(B, N, F, device=dev, generator=g).to(torch.bfloat16)
rna = torch.randn(B, 2048, device=dev, generator=g)
kw = {}
if c4:
    lens = torch.randint(2048, N + 1, (B,), device=dev, generator=g)
    kw["wsi_key_padding_mask"] = torch.arange(N, device=dev)[None, :] < lens[:, None]
for _ in range(6):
    eng.step(wsi, rna, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    eng.step(wsi, rna, **kw)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / steps * 1e3:.3f}")
