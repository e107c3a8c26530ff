#!/usr/bin/env python3
"""Does a pinned host -> device copy on a side stream overlap kernels on the main stream on this box?"""
import time, torch
dev = torch.device("cuda", 0)
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
h = torch.empty(134 * 1024 * 1024 // 2, dtype=torch.bfloat16).pin_memory()
d = torch.empty_like(h, device=dev)
side = torch.cuda.Stream(device=dev)
def work(n):
    for _ in range(n):
        torch.mm(a, a)
def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
work(3)
tw = t(lambda: work(10))
tc = t(lambda: d.copy_(h, non_blocking=True))
def both():
    with torch.cuda.stream(side):
        d.copy_(h, non_blocking=True)
    work(10)
tb = t(both)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    work(10)
def both_graph():
    with torch.cuda.stream(side):
        d.copy_(h, non_blocking=True)
    g.replay()
tg = t(lambda: g.replay())
tbg = t(both_graph)
print(f"work {tw:.2f} ms, copy {tc:.2f} ms, both (eager) {tb:.2f} ms, graph {tg:.2f} ms, both (graph) {tbg:.2f} ms")
