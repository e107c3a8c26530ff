"""Which host lines launch the torch-side kernels of a step (copies, adds, fills, RNG)?  One eager engine step under torch.profiler
with python stacks; every top-level aten op that launched a device kernel is attributed to its innermost mirror_amd frame.
Usage (GPU box): python tools/exp/attribute_torch_launches.py > gpurun_out/attr.txt"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import CONFIGS  # noqa: E402
import mirror_amd.models as M  # noqa: E402
from mirror_amd import functional as Fn  # noqa: E402
from mirror_amd.engine import TrainEngine  # noqa: E402
from mirror_amd.losses import MIRRORLoss  # noqa: E402

dev = torch.device("cuda", 0)
shp = CONFIGS["c2"]
torch.manual_seed(42)
model = M.mirror(wsi_embed_dim=shp["F"], rna_embed_dim=shp["G"], embed_dim=shp["D"], wsi_num_tokens=shp["N"],
                 rna_encoder_depth=shp["L"], rna_mlp_ratio=shp["mlp"], rna_norm_layer="layernorm", rna_act_layer="gelu",
                 rna_num_heads=shp["heads"]).to(dev).train()
loss_fn = MIRRORLoss(alignment_loss_weight=0.5, wsi_retention_loss_weight=0.15, rna_retention_loss_weight=0.15,
                     style_loss_weight=0.1, cluster_loss_weight=0.1, gather_distributed=False)
eng = TrainEngine(model, loss_fn, lr=2e-5, precision="bf16")
Fn.manual_seed(1234)
g = torch.Generator(device=dev).manual_seed(1234)
wsi = torch.randn(16, shp["N"], shp["F"], device=dev, generator=g).to(torch.bfloat16)
rna = torch.randn(16, shp["G"], device=dev, generator=g)
eng.step(wsi, rna)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    eng.step(wsi, rna)
    torch.cuda.synchronize()

def own_frame(stack):
    for fr in stack:
        if "mirror_amd/" in fr:
            return fr.split("mirror_amd/")[-1]
    return "(no mirror_amd frame: autograd engine)"

cnt = collections.Counter()
ker = collections.Counter()
for e in prof.events():
    if not e.name.startswith("aten::"):
        continue
    p = e.cpu_parent
    if p is not None and p.name.startswith("aten::"):
        continue                      # only the outermost aten op
    nk = 0
    todo = [e]
    while todo:
        x = todo.pop()
        nk += len(x.kernels)
        todo.extend(x.cpu_children)
    if nk == 0:
        continue
    parent = p.name if p is not None else "-"
    cnt[(e.name, own_frame(e.stack), parent[:60])] += nk
    ker[e.name] += nk
print("device launches under aten ops, by op:")
for k, v in ker.most_common():
    print(f"  {v:4d}  {k}")
print("\nby op, innermost mirror_amd frame, enclosing profiler range:")
for (op, fr, par), v in sorted(cnt.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"  {v:3d}  {op:22s} {fr:70s} {par}")
