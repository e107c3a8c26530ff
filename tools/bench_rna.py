#!/usr/bin/env python3
"""The RNA branch (c2: G = 2048, D = 512, depth 6 + 1 decoder block, B = 16) alone on the chip: forward + backward time and
launch count, fused Block calls vs the composed ops (MIRROR_RNA_FUSED)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mirror_amd.models as M
from mirror_amd import functional as Fn

dev = torch.device("cuda", 0)
torch.manual_seed(1)
m = M.mirror(wsi_embed_dim=64, rna_embed_dim=2048, embed_dim=512, wsi_num_tokens=16, rna_encoder_depth=6, rna_mlp_ratio=4.0,
             rna_num_heads=8, num_prototypes=10).to(dev).train()
m.precision = "bf16"
rna = torch.randn(16, 2048, device=dev)
noise = torch.rand(16, 512, device=dev)
for fused in (True, False, True):
    Fn._RNA_FUSED = fused
    def step():
        Fn._dropout_state["offset"] = 0
        outs = m.rna_branch(rna, noise, 0.75)
        loss = sum(o.float().sum() for o in outs[:3])
        loss.backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print(f"fused={fused}: graph replay {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per fwd+bwd")
