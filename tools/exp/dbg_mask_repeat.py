#!/usr/bin/env python3
"""Two identical eager masked steps (fresh engines, same seeds): which parameters' gradients are not reproducible?  --off hook,hook"""
import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
import test_engine_gpu as T
import mirror_amd.models as M
from mirror_amd.engine import TrainEngine
from mirror_amd.losses import MIRRORLoss
from mirror_amd import functional as Fn, kernels as K
for a in sys.argv[1:]:
    name, val = a.split("=")
    mod, attr = name.rsplit(".", 1)
    m = importlib.import_module("mirror_amd." + mod)
    assert hasattr(m, attr), name
    setattr(m, attr, eval(val))
CFG512 = T.CFG512
n = CFG512["wsi_num_tokens"]
masked = os.environ.get("MASKED", "1") == "1"
snaps = []
for rep in range(int(os.environ.get("REPS", "4"))):
    torch.manual_seed(21)
    model = M.mirror(**CFG512, rna_proj_drop_rate=0.1).cuda().train()
    eng = TrainEngine(model, MIRRORLoss(), lr=1e-6, precision="bf16", graph=False, seed=77, snapshot_grads=True)
    eng._rna_branch_state = "off"
    wsi, rna, _ = T._batch(4, 5, CFG512)
    lens = torch.tensor([n, 700, 333, 512], device="cuda")
    mask = torch.arange(n, device="cuda")[None, :] < lens[:, None]
    wsi = (wsi * mask[..., None]).to(torch.bfloat16) if masked else wsi.to(torch.bfloat16)
    if os.environ.get("LENS2"):
        mask = torch.arange(n, device="cuda")[None, :] < torch.tensor([600, n, 400, 900], device="cuda")[:, None]
    for _ in range(int(os.environ.get("NSTEP", "1"))):
        l = eng.step(wsi, rna, **({"wsi_key_padding_mask": mask} if masked else {}))
    torch.cuda.synchronize()
    names = {id(p): k for k, p in model.named_parameters()}
    snaps.append((eng.grad_snap.clone(), [(names[id(p)], o, p.numel()) for p, o in zip(eng.params, eng.offsets)]))
g0, lay = snaps[0]
for r, (g, _) in enumerate(snaps[1:], 1):
    rows = []
    for k, o, m in lay:
        a, b = g0[o:o + m], g[o:o + m]
        rows.append((float((a - b).norm()) / max(float(a.norm()), 1e-12), k))
    rows.sort(reverse=True)
    print(f"rep {r} vs 0: total rel {float((g - g0).norm() / g0.norm()):.2e}; worst:", ", ".join(f"{k} {d:.1e}" for d, k in rows[:3]))
