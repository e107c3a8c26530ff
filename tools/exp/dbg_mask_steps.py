#!/usr/bin/env python3
"""Per-step reproducibility of the masked eager step: two identical engines, the gradient of _fc1.bias and the whole arena after each step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
import test_engine_gpu as T
import mirror_amd.models as M
from mirror_amd.engine import TrainEngine
from mirror_amd.losses import MIRRORLoss
CFG512 = T.CFG512
n = CFG512["wsi_num_tokens"]
runs = []
for rep in range(3):
    torch.manual_seed(21)
    model = M.mirror(**CFG512, rna_proj_drop_rate=0.1).cuda().train()
    eng = TrainEngine(model, MIRRORLoss(), lr=1e-6, precision="bf16", graph=False, seed=77, snapshot_grads=True)
    eng._rna_branch_state = "off"
    wsi, rna, _ = T._batch(4, 5, CFG512)
    lens = torch.tensor([n, 700, 333, 512], device="cuda")
    mask = torch.arange(n, device="cuda")[None, :] < lens[:, None]
    wsi = (wsi * mask[..., None]).to(torch.bfloat16)
    names = {id(p): k for k, p in model.named_parameters()}
    o = [o for p, o in zip(eng.params, eng.offsets) if names[id(p)] == "wsi_encoder._fc1.0.bias"][0]
    per = []
    for s in range(7):
        if s == 5:
            mask.copy_(torch.arange(n, device="cuda")[None, :] < torch.tensor([600, n, 400, 900], device="cuda")[:, None])
        l = [float(x) for x in eng.step(wsi, rna, wsi_key_padding_mask=mask)]
        per.append((eng.grad_snap.clone(), eng.grad_snap[o:o + 512].clone(), l, eng.master[o:o + 512].clone()))
    runs.append(per)
for r in (1, 2):
    print(f"rep {r} vs 0:")
    for s in range(7):
        g0, b0, l0, p0 = runs[0][s]
        g1, b1, l1, p1 = runs[r][s]
        d = (b1 - b0)
        print(f"  step {s}: arena rel {float((g1 - g0).norm() / g0.norm()):.2e}  fc1.bias rel {float(d.norm() / b0.norm()):.2e}  max|d| at ch {int(d.abs().argmax())} = {float(d.abs().max()):.3e}"
              f"  (|d| > 0.1 max: {int((d.abs() > 0.1 * d.abs().max()).sum())} channels)  bias there {float(p0[int(d.abs().argmax())]):+.3e}  loss {l0[0]:.5f}/{l1[0]:.5f}")
