#!/usr/bin/env python3
"""Where the branches of the replayed c2 step start and end ON THE DEVICE, without a tracer attached: one-thread launches that
store the device wall clock at named points of the step (MIRROR_PROBE=1 -> functional.probe / ProbeFn -> mh_timestamp).
usage (GPU box, repo root): MIRROR_PROBE=1 python3 tools/exp/probe_timeline.py [--steps 12]"""
import argparse
import os
import sys

os.environ.setdefault("MIRROR_PROBE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import mirror_amd.models as M                     # noqa: E402
from mirror_amd.engine import TrainEngine         # noqa: E402
from mirror_amd.losses import MIRRORLoss          # noqa: E402
from mirror_amd import functional as Fn           # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--pipelined", type=int, default=1, help="1: the timed steps are issued back to back (as bench.py does) and the last is read")
ap.add_argument("--on", default="", help="comma-separated hooks to turn on")
ap.add_argument("--off", default="", help="comma-separated test hooks of mirror_amd.functional to turn off (e.g. _FAN_IN_LN_BWD)")
a = ap.parse_args()
for _name in filter(None, a.on.split(",")):
    assert hasattr(Fn, _name), _name
    setattr(Fn, _name, True)
for _name in filter(None, a.off.split(",")):
    assert hasattr(Fn, _name), _name
    setattr(Fn, _name, False)
dev = torch.device("cuda", 0)
torch.manual_seed(42)
model = M.mirror(wsi_embed_dim=1024, rna_embed_dim=2048, embed_dim=512, wsi_num_tokens=4096, rna_encoder_depth=6, rna_mlp_ratio=4.0,
                 rna_norm_layer="layernorm", rna_act_layer="gelu", rna_num_heads=8).to(dev).train()
eng = TrainEngine(model, MIRRORLoss(alignment_loss_weight=0.5, wsi_retention_loss_weight=0.15, rna_retention_loss_weight=0.15,
                                    style_loss_weight=0.1, cluster_loss_weight=0.1), lr=2e-5, precision="bf16")
Fn.manual_seed(1234)
g = torch.Generator(device=dev).manual_seed(1234)
wsi = torch.randn(16, 4096, 1024, device=dev, generator=g).to(torch.bfloat16)
rna = torch.randn(16, 2048, device=dev, generator=g)
for _ in range(6):
    eng.step(wsi, rna)
acc = {}
prev_start = None
per = []
for _ in range(a.steps):
    for _ in range(4 if a.pipelined else 1):      # back to back: the host runs ahead as in bench.py; the probes hold the last one
        eng.step(wsi, rna)
    t = Fn.probe_read()
    t0 = t["step_start"]
    if prev_start is not None:
        per.append((t0 - prev_start) / 100.0)
    prev_start = t0
    for k, v in t.items():
        acc.setdefault(k, []).append((v - t0) / 100.0)      # 100 MHz ticks -> us
print(f"mean distance between read steps: {sum(per) / max(len(per), 1):.1f} us (4 steps when pipelined)")
for k, v in sorted(acc.items(), key=lambda kv: sum(kv[1]) / len(kv[1])):
    v = sorted(v)
    print(f"{sum(v) / len(v):9.1f} us  (min {v[0]:8.1f} max {v[-1]:8.1f})  {k}")
