#!/usr/bin/env python3
"""GPU: loss / gradient error of each precision policy against the reference golden vectors (reported, not gated)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.golden_util import ModelCase, TEMPLATE_W  # noqa: E402
from tests.test_model_gpu import build, run, W_KW  # noqa: E402
from mirror_amd.losses import MIRRORLoss  # noqa: E402

for name in sys.argv[1:] or ["c1"]:
    case = ModelCase(name)
    for prec in ("fp32", "bf16_pinv32", "bf16"):
        model = build(case, precision=prec)
        outs = run(case, model)
        loss = MIRRORLoss(**dict(zip(W_KW, TEMPLATE_W)))(*outs)
        loss[0].backward()
        got = np.array([float(x.detach()) for x in loss])
        ref = case.z["loss_template"]
        params = dict(model.named_parameters())
        gn = np.array([float(params[k].grad.double().norm()) for k in case.keys])
        gref = case.z["grad_norm"]
        rel = np.abs(gn - gref) / np.maximum(gref, 1e-12)
        worst = np.argsort(-rel)[:3]
        emb = max(float((a.detach().float().cpu().flatten()[torch.from_numpy(case.z[f'out_idx/{nm}'])] - torch.from_numpy(case.z[f'out_val/{nm}'])).abs().max()) / max(float(np.abs(case.z[f'out_val/{nm}']).max()), 1e-6)
                  for nm, a in zip(("wsi_alignment_emb", "rna_alignment_emb", "wsi_retention_emb"), (outs[0], outs[7], outs[1])))
        print(f"{name} {prec:9s} loss rel err {np.abs(got - ref) / np.abs(ref)}  emb max-abs/max {emb:.2e}  "
              f"grad-norm rel err median {np.median(rel):.2e} max {rel.max():.2e} ({', '.join(case.keys[i] for i in worst)})")
