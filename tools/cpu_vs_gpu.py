#!/usr/bin/env python3
"""Is the step CPU-bound?  Time the host-side enqueue of K steps (no sync inside) against the GPU completion time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mirror_amd.models as M
from mirror_amd.engine import TrainEngine
from mirror_amd.losses import MIRRORLoss
from mirror_amd import functional as Fn

dev = torch.device("cuda", 0)
torch.manual_seed(42)
model = M.mirror(wsi_embed_dim=1024, rna_embed_dim=2048, embed_dim=512, wsi_num_tokens=4096, rna_encoder_depth=6,
                 rna_mlp_ratio=4.0, rna_norm_layer="layernorm", rna_act_layer="gelu", rna_num_heads=8).to(dev).train()
loss_fn = MIRRORLoss(alignment_loss_weight=0.5, wsi_retention_loss_weight=0.15, rna_retention_loss_weight=0.15,
                     style_loss_weight=0.1, cluster_loss_weight=0.1)
eng = TrainEngine(model, loss_fn, lr=2e-5, precision="bf16")
Fn.manual_seed(1234)
wsi = torch.randn(16, 4096, 1024, device=dev).to(torch.bfloat16)
rna = torch.randn(16, 2048, device=dev)
for _ in range(5):
    eng.step(wsi, rna)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    eng.step(wsi, rna)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / K:.2f} ms/step; until GPU done {1e3 * (t2 - t0) / K:.2f} ms/step")

if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        eng.step(wsi, rna)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
