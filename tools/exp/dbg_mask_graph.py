#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch
import test_engine_gpu as T
import mirror_amd.models as M
from mirror_amd.engine import TrainEngine
from mirror_amd.losses import MIRRORLoss
from mirror_amd import functional as Fn
for sw in sys.argv[1:]:
    setattr(Fn, sw, False)
CFG512 = T.CFG512
def once():
    n = CFG512["wsi_num_tokens"]
    runs = []
    for graph in [bool(int(x)) for x in os.environ.get('MODES', '1,0').split(',')]:
        torch.manual_seed(21)
        model = M.mirror(**CFG512, rna_proj_drop_rate=0.1).cuda().train()
        eng = TrainEngine(model, MIRRORLoss(), lr=float(os.environ.get("LR", "1e-4")), precision="bf16", graph=graph, seed=77, snapshot_grads=True)
        if not graph:
            eng._rna_branch_state = "off"
        wsi, rna, _ = T._batch(4, 5, CFG512)
        lens = torch.tensor([n, 700, 333, 512], device="cuda")
        mask = torch.arange(n, device="cuda")[None, :] < lens[:, None]
        wsi = (wsi * mask[..., None]).to(torch.bfloat16)
        torch.manual_seed(123)
        for _ in range(int(os.environ.get("NSTEP", "5"))):
            l = [float(x) for x in eng.step(wsi, rna, wsi_key_padding_mask=mask)]
        if os.environ.get("MASK2"):
            mask.copy_(torch.arange(n, device="cuda")[None, :] < torch.tensor([600, n, 400, 900], device="cuda")[:, None])
            l = [float(x) for x in eng.step(wsi, rna, wsi_key_padding_mask=mask)]
        print("loss", [round(float(x), 5) for x in l])
        snap = eng.grad_snap.clone()
        names = {id(p): k for k, p in model.named_parameters()}
        if graph and os.environ.get("EXTRA"):
            eng.step(wsi, rna)
        runs.append((snap, [(names[id(p)], o, p.numel()) for p, o in zip(eng.params, eng.offsets)]))
    (ga, lay), (gb, _) = runs
    print("total rel", float((ga - gb).norm() / gb.norm()))
    rows = []
    for k, o, m in lay:
        a, b = ga[o:o + m], gb[o:o + m]
        d = float((a - b).norm())
        rows.append((d, k, float(b.norm()), float(a.norm())))
    rows.sort(reverse=True)
    for d, k, nb, na in rows[:4]:
        print(f"{k:60s} diff {d:.4e} |eager| {nb:.4e} |graph| {na:.4e}")

for it in range(int(os.environ.get('ITERS', '1'))):
    once()
