#!/usr/bin/env python3
"""H2D copy beside the REAL replayed c2 step (row f2).  (a) replay alone; (b) an independent pinned -> device copy on a side stream beside
each replay (no event anywhere); (c) the same with the copy's stream waiting for an event recorded behind the PREVIOUS replay (what
HostFeeder's `free` does); (d) + the main stream waiting for the copy made during the previous replay (HostFeeder's `ready`)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mirror_amd.models as M
from mirror_amd.engine import TrainEngine
from mirror_amd.losses import MIRRORLoss
dev = torch.device("cuda", 0)
B, N, F, G, D, L = 16, 4096, 1024, 2048, 512, 6
torch.manual_seed(42)
model = M.mirror(wsi_embed_dim=F, rna_embed_dim=G, embed_dim=D, wsi_num_tokens=N, rna_encoder_depth=L, rna_mlp_ratio=4.0, rna_num_heads=8).to(dev).train()
eng = TrainEngine(model, MIRRORLoss(), lr=2e-5, precision="bf16")
g = torch.Generator(device=dev).manual_seed(1)
wsi = torch.randn(B, N, F, device=dev, generator=g).to(torch.bfloat16)
rna = torch.randn(B, G, device=dev, generator=g)
for _ in range(5):
    eng.step(wsi, rna)
torch.cuda.synchronize()
assert eng._graph is not None
h = [torch.empty(B * N * F, dtype=torch.bfloat16).pin_memory() for _ in range(2)]
d = [torch.empty(B * N * F, dtype=torch.bfloat16, device=dev) for _ in range(2)]
side = torch.cuda.Stream(device=dev, priority=int(os.environ.get("SIDE_PRIO", "0")))
print("side stream priority", side.priority, "GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
def run(mode, steps=12):
    torch.cuda.synchronize()
    free = [None, None]
    ready = [None, None]
    t0 = time.perf_counter()
    for j in range(steps):
        s = j % 2
        if mode >= 1:
            with torch.cuda.stream(side):
                if mode >= 2 and free[s] is not None:
                    side.wait_event(free[s])
                d[s].copy_(h[s], non_blocking=True)
                ready[s] = side.record_event()
        if mode >= 3 and ready[1 - s] is not None:
            torch.cuda.current_stream().wait_event(ready[1 - s])
        eng.step(wsi, rna)
        free[1 - s] = torch.cuda.current_stream().record_event()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps
def run_host_wait(steps=12, depth=3, device_wait=False):
    """the copy's slot is protected by the event of the step `depth` steps back, waited for on the HOST (long complete: no block) or,
    device_wait=True, by a stream wait on that same (already complete) event"""
    torch.cuda.synchronize()
    hh = [torch.empty(B * N * F, dtype=torch.bfloat16).pin_memory() for _ in range(depth)]
    dd = [torch.empty(B * N * F, dtype=torch.bfloat16, device=dev) for _ in range(depth)]
    free = [None] * depth
    ready = [None] * depth
    t0 = time.perf_counter()
    for j in range(steps):
        s = j % depth
        if free[s] is not None:
            if device_wait:
                side.wait_event(free[s])
            else:
                free[s].synchronize()
        with torch.cuda.stream(side):
            dd[s].copy_(hh[s], non_blocking=True)
            ready[s] = side.record_event()
        p = (j - 1) % depth
        if ready[p] is not None and j > 0:
            torch.cuda.current_stream().wait_event(ready[p])
        eng.step(wsi, rna)
        free[p] = torch.cuda.current_stream().record_event()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps
print(f"{'3 slots, HOST wait on the old event, replay waits for the copy':50s} {run_host_wait():7.3f} ms/step", flush=True)
print(f"{'3 slots, DEVICE wait on the old (complete) event':50s} {run_host_wait(device_wait=True):7.3f} ms/step", flush=True)
for mode, name in ((0, "replay alone"), (1, "+ independent copy beside every replay"), (2, "+ copy waits for the previous replay's event"),
                   (3, "+ replay waits for the previous copy"), (0, "replay alone (again)")):
    print(f"{name:50s} {run(mode):7.3f} ms/step", flush=True)
