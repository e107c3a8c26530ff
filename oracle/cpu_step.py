"""CPU baseline step (forward + MIRRORLoss + backward + Adam) on the oracle.  TEST/BENCH INFRASTRUCTURE:
used only by bench.py's `cpu_baseline` leg to time the reference algorithm on the GPU box's host cores."""
from __future__ import annotations

import time
from typing import Dict

import torch

from . import mirror_oracle as O
from . import synth


def time_cpu_steps(cfg: O.Cfg, batch: int, steps: int, warmup: int = 1, seed: int = 1234, threads: int = 0) -> Dict:
    if threads:
        torch.set_num_threads(threads)
    sd = {k: v.clone().requires_grad_(True) for k, v in synth.synth_state_dict(synth.param_shapes(cfg), seed).items()}
    opt = torch.optim.Adam(list(sd.values()), lr=2e-5)
    wsi, rna, noise = synth.synth_batch(cfg, batch, seed + 1)
    weights = (0.5, 0.15, 0.15, 0.1, 0.1)
    times = []
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        with torch.no_grad():
            sd["prototypes.weight"].copy_(torch.nn.functional.normalize(sd["prototypes.weight"], dim=1))
        outs = O.mirror_forward(sd, cfg, wsi, rna, noise)
        loss = O.mirror_loss(outs, weights)[0]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        if i >= warmup:
            times.append(time.perf_counter() - t0)
    per_step = sum(times) / len(times)
    return {"samples_per_s": batch / per_step, "s_per_step": per_step, "cores": torch.get_num_threads(),
            "steps": steps, "batch": batch, "loss": float(loss.detach())}
