#!/usr/bin/env python3
"""BASELINE config 4 at full size: Phikon-dim features (768-d), 8192 patch tokens per slide, per-sample valid length
~U[2048, 8192] (seeded), padded + bool key-padding mask through every Nystrom layer.  Prints one JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mirror_amd.models as M
from mirror_amd.engine import TrainEngine
from mirror_amd.losses import MIRRORLoss

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--precision", default="bf16")
a = ap.parse_args()
dev = torch.device("cuda", 0)
N, F, G, D, L = 8192, 768, 2048, 512, 6
torch.manual_seed(42)
model = M.mirror(wsi_embed_dim=F, rna_embed_dim=G, embed_dim=D, wsi_num_tokens=N, rna_encoder_depth=L, rna_mlp_ratio=4.0,
                 rna_norm_layer="layernorm", rna_act_layer="gelu", rna_num_heads=8).to(dev).train()
eng = TrainEngine(model, MIRRORLoss(alignment_loss_weight=0.5, wsi_retention_loss_weight=0.15, rna_retention_loss_weight=0.15,
                                    style_loss_weight=0.1, cluster_loss_weight=0.1), lr=2e-5, precision=a.precision)
g = torch.Generator(device=dev).manual_seed(1234)
wsi = torch.randn(a.batch, N, F, device=dev, generator=g).to(torch.float32 if a.precision == "fp32" else torch.bfloat16)
rna = torch.randn(a.batch, G, device=dev, generator=g)
lens = torch.randint(2048, N + 1, (a.batch,), device=dev, generator=g)
mask = torch.arange(N, device=dev)[None, :] < lens[:, None]
wsi = wsi * mask[..., None]                                   # padded rows are zeros, as a collate function would leave them
for _ in range(4):      # two warm steps, the capture, one replayed step
    losses = eng.step(wsi, rna, wsi_key_padding_mask=mask)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    losses = eng.step(wsi, rna, wsi_key_padding_mask=mask)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"config": "BASELINE c4: 8192 x 768-d patch tokens, valid length ~U[2048, 8192], key-padding mask", "precision": a.precision,
                  "batch": a.batch, "steps": a.steps, "ms_per_step": round(dt / a.steps * 1e3, 2), "samples_per_s": round(a.batch * a.steps / dt, 2),
                  "step_launch": "hip_graph" if eng._graph is not None else "eager", "valid_lengths": lens.tolist(), "losses": [round(float(x), 5) for x in losses],
                  "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))
