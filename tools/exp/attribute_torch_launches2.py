"""Second pass of attribute_torch_launches.py: (a) which parameters still get their gradient through AccumulateGrad (one torch add_
each) instead of the arena sink, (b) which host lines call the torch fills / RNG / copies.  One eager engine step.
Usage (GPU box): python tools/exp/attribute_torch_launches2.py > gpurun_out/attr2.txt"""
import collections
import os
import sys
import traceback

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

hits = collections.Counter()
names = {p: n for n, p in model.named_parameters()}
for p, n in names.items():
    p.register_hook(lambda gr, n=n: (hits.update([n]) if gr is not None else None) and None)

calls = collections.Counter()


def frame():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "mirror_amd/" in fr.filename:
            return f"{fr.filename.split('mirror_amd/')[-1]}:{fr.lineno} {fr.name}"
    return "?"


def wrap(obj, name):
    orig = getattr(obj, name)

    def f(*a, **k):
        calls[(name, frame())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)


for nm in ("zero_", "add_", "add", "copy_", "clone", "fill_", "exp", "mul", "__add__", "__iadd__", "__mul__", "__setitem__", "sum", "mean"):
    wrap(torch.Tensor, nm)
for nm in ("zeros", "zeros_like", "rand", "randn", "stack", "cat", "ones", "full"):
    wrap(torch, nm)
eng.step(wsi, rna)
torch.cuda.synchronize()
print("parameters whose gradient went through AccumulateGrad this step:")
for n, c in hits.most_common():
    print(f"  {c}  {n}  {tuple(dict(model.named_parameters())[n].shape)}")
print("\npython-level torch calls inside mirror_amd during the step (not all launch a kernel):")
for (nm, fr), c in sorted(calls.items(), key=lambda kv: (kv[0][0], -kv[1])):
    if nm in ("empty_like", "to", "float") and c < 1:
        continue
    print(f"  {c:3d}  {nm:12s} {fr}")
