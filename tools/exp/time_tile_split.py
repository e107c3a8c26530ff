#!/usr/bin/env python3
"""A chain of dependent 384-cubed tile products over B h = 128 matrices (the template's Moore-Penrose iteration) as ONE chain of
256-workgroup launches against TWO (FOUR) concurrent chains over halves (quarters) of the batch on as many streams: us per product."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mirror_amd import kernels as K  # noqa: E402
from mirror_amd._lib import MH_BF16  # noqa: E402
dev = torch.device("cuda", 0)
B, h, m, n = 16, 8, 384, 24
x = (torch.randn(B, h, m, m, device=dev) * 0.05).bfloat16()
bufs = [torch.empty_like(x) for _ in range(3)]


def chain(lo, hi):
    z = x[lo:hi]
    for i in range(n):
        out = bufs[i % 3][lo:hi]
        K.gemm(x[lo:hi], z, out, alpha=0.25, mma=MH_BF16)
        z = out


def run(parts):
    cur = torch.cuda.current_stream()
    if parts == 1:
        chain(0, B)
        return
    step = B // parts
    for i in range(parts):
        s = streams[i]
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            chain(i * step, (i + 1) * step)
    for i in range(parts):
        cur.wait_stream(streams[i])


streams = [torch.cuda.Stream() for _ in range(4)]
for parts in (1, 2, 4, 1, 2, 4):
    g = torch.cuda.CUDAGraph()
    run(parts); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        run(parts)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{parts} chain(s): {e0.elapsed_time(e1) / 10 / n * 1e3:7.1f} us per product step ({n} dependent products, graph replay)", flush=True)
