"""CPU baseline step (forward + MIRRORLoss + backward + Adam) on the oracle.  TEST/BENCH INFRASTRUCTURE:
used only by bench.py's `cpu_baseline` leg to time the reference algorithm on the GPU box's host cores."""
from __future__ import annotations

import os
import time
from typing import Dict

import torch

from . import mirror_oracle as O
from . import synth


def usable_cores(cap: int = 32) -> int:
    """Cores this process may run on (cgroup/affinity aware), capped: torch's CPU GEMMs stop scaling (and with 256
    logical CPUs reported but far fewer granted, collapse) long before that."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:  # cgroup v2 CPU quota
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


def time_cpu_steps(cfg: O.Cfg, batch: int, budget_s: float = 25.0, seed: int = 1234, threads: int = 0) -> Dict:
    """Bounded sample: steps are timed until `budget_s` of CPU wall time is spent (the first step counts when it
    alone already exceeds half the budget)."""
    torch.set_num_threads(threads or usable_cores())
    sd = {k: v.clone().requires_grad_(True) for k, v in synth.synth_state_dict(synth.param_shapes(cfg), seed).items()}
    opt = torch.optim.Adam(list(sd.values()), lr=2e-5)
    wsi, rna, noise = synth.synth_batch(cfg, batch, seed + 1)
    weights = (0.5, 0.15, 0.15, 0.1, 0.1)
    times = []
    t_start = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        with torch.no_grad():
            sd["prototypes.weight"].copy_(torch.nn.functional.normalize(sd["prototypes.weight"], dim=1))
        outs = O.mirror_forward(sd, cfg, wsi, rna, noise)
        loss = O.mirror_loss(outs, weights)[0]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s or len(times) >= 6:
            break
    timed = times[1:] if len(times) > 1 else times     # drop the warm-up step when there is more than one
    per_step = sum(timed) / len(timed)
    return {"samples_per_s": batch / per_step, "s_per_step": per_step, "cores": torch.get_num_threads(),
            "steps": len(timed), "warmup": len(times) - len(timed), "batch": batch, "loss": float(loss.detach())}
