#!/usr/bin/env python3
"""cProfile of the eager training step's host side (what bounds the multi-GPU path, which does not replay a HIP graph)."""
import cProfile, os, pstats, sys, time
os.environ["MIRROR_GRAPH"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
t0 = time.perf_counter()
for _ in range(5):
    eng.step(wsi, rna)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / 5:.2f} ms/step, with drain {1e3 * (t2 - t0) / 5:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    eng.step(wsi, rna)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
