#!/usr/bin/env python3
"""Poison every torch.empty / empty_like float buffer with NaN and run one masked (or unmasked) eager step: a NaN in a loss or gradient
means some kernel read rows nobody wrote.  POISON=nan|big|off  MASKED=0|1"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
_empty, _empty_like = torch.empty, torch.empty_like
mode = os.environ.get("POISON", "nan")
val = float("nan") if mode == "nan" else 1e4
log = []


def _poison(t):
    if mode != "off" and t.is_cuda and t.is_floating_point() and t.numel():
        t.fill_(val)
    return t


def empty(*a, **kw):
    return _poison(_empty(*a, **kw))


def empty_like(*a, **kw):
    return _poison(_empty_like(*a, **kw))


torch.empty, torch.empty_like = empty, empty_like
import test_engine_gpu as T
import mirror_amd.models as M
from mirror_amd.engine import TrainEngine
from mirror_amd.losses import MIRRORLoss
CFG512 = T.CFG512
n = CFG512["wsi_num_tokens"]
masked = os.environ.get("MASKED", "1") == "1"
torch.manual_seed(21)
model = M.mirror(**CFG512, rna_proj_drop_rate=0.1).cuda().train()
eng = TrainEngine(model, MIRRORLoss(), lr=1e-6, precision="bf16", graph=False, seed=77, snapshot_grads=True)
eng._rna_branch_state = "off"
wsi, rna, _ = T._batch(4, 5, CFG512)
lens = torch.tensor([600, n, 400, 900], device="cuda")
mask = torch.arange(n, device="cuda")[None, :] < lens[:, None]
wsi = wsi.to(torch.bfloat16)          # NOT zeroed at the padded rows: a padded row's features must not matter
l = eng.step(wsi, rna, **({"wsi_key_padding_mask": mask} if masked else {}))
torch.cuda.synchronize()
print("losses", [float(x) for x in l])
names = {id(p): k for k, p in model.named_parameters()}
bad = []
for p, o in zip(eng.params, eng.offsets):
    g = eng.grad_snap[o:o + p.numel()]
    if not bool(torch.isfinite(g).all()) or float(g.abs().max()) > 1e3:
        bad.append(names[id(p)])
print(len(bad), "parameters with non-finite / huge gradients:", bad[:12])
