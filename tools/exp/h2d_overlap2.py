#!/usr/bin/env python3
"""Where does a pinned host -> device copy go relative to a replayed HIP graph?  (row f2: --feed host-bf16 runs 22 % behind the resident
batch.)  Variants: copy enqueued BEFORE / AFTER the replay on a side stream; the copy as a node of the graph itself (side branch);
the copy split into pieces; HSA_ENABLE_SDMA / queue-count settings come from the environment of the call."""
import os, time, torch
dev = torch.device("cuda", 0)
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
NB = 134 * 1024 * 1024
h = torch.empty(NB // 2, dtype=torch.bfloat16).pin_memory()
d = torch.empty_like(h, device=dev)
side = torch.cuda.Stream(device=dev)
def work(n):
    for _ in range(n):
        torch.mm(a, a)
def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / reps
work(3)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    work(8)
tg = t(lambda: g.replay())
tc = t(lambda: d.copy_(h, non_blocking=True))
def copy_then_graph():
    with torch.cuda.stream(side):
        d.copy_(h, non_blocking=True)
    g.replay()
def graph_then_copy():
    g.replay()
    with torch.cuda.stream(side):
        d.copy_(h, non_blocking=True)
def pieces_then_graph(n=16):
    with torch.cuda.stream(side):
        m = h.numel() // n
        for i in range(n):
            d[i * m:(i + 1) * m].copy_(h[i * m:(i + 1) * m], non_blocking=True)
    g.replay()
# the copy as a node of the graph, on a branch that nothing waits for until the end
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    main = torch.cuda.current_stream()
    side2 = torch.cuda.Stream(device=dev)
    side2.wait_stream(main)
    with torch.cuda.stream(side2):
        d.copy_(h, non_blocking=True)
    work(8)
    main.wait_stream(side2)
res = {"graph": tg, "copy": tc, "copy_then_graph": t(copy_then_graph), "graph_then_copy": t(graph_then_copy),
       "16 pieces then graph": t(pieces_then_graph), "copy inside graph": t(lambda: g2.replay())}
print(" | ".join(f"{k} {v:.2f} ms" for k, v in res.items()), " env:", {k: os.environ.get(k) for k in ("HSA_ENABLE_SDMA", "GPU_MAX_HW_QUEUES")}, flush=True)
