#!/usr/bin/env python3
"""Print every K.gemm call of one eager training step at the bench workload: shapes, split, strides."""
import os, sys, collections
os.environ["MIRROR_GRAPH"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirror_amd import kernels as K

calls = []
orig = K.gemm
def spy(a, b, *args, **kw):
    calls.append((tuple(a.shape), tuple(a.stride()), tuple(b.shape), tuple(b.stride()), kw.get("split_k", 1), bool(kw.get("accumulate", False)),
                  str(kw.get("out_dtype") or (kw["out"].dtype if kw.get("out") is not None else None))))
    return orig(a, b, *args, **kw)
K.gemm = spy
import mirror_amd.functional as Fn
Fn.K.gemm = spy
import mirror_amd.models as M
from mirror_amd.engine import TrainEngine
from mirror_amd.losses import MIRRORLoss
dev = torch.device("cuda", 0)
torch.manual_seed(42)
model = M.mirror(wsi_embed_dim=1024, rna_embed_dim=2048, embed_dim=512, wsi_num_tokens=4096, rna_encoder_depth=6, rna_mlp_ratio=4.0,
                 rna_norm_layer="layernorm", rna_act_layer="gelu", rna_num_heads=8).to(dev).train()
loss_fn = MIRRORLoss(alignment_loss_weight=0.5, wsi_retention_loss_weight=0.15, rna_retention_loss_weight=0.15,
                     style_loss_weight=0.1, cluster_loss_weight=0.1, gather_distributed=False)
eng = TrainEngine(model, loss_fn, lr=2e-5, precision="bf16")
wsi = torch.randn(16, 4096, 1024, device=dev).bfloat16()
rna = torch.randn(16, 2048, device=dev)
for _ in range(2):
    eng.step(wsi, rna)
calls.clear()
eng.step(wsi, rna)
torch.cuda.synchronize()
for c in calls:
    print(c)
