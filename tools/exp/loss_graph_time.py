#!/usr/bin/env python3
"""Untraced cost of the loss section (MIRRORLoss forward + backward down to its input gradients) as a HIP-graph replay:
how much of the forward / backward boundary of the step is launch latency of ~45 tiny kernels?"""
import torch
from mirror_amd.losses import MIRRORLoss

dev, bf = "cuda", torch.bfloat16
B, N, F, G, D, P, S = 16, 4096, 1024, 2048, 512, 3000, 128
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s, dt=torch.float32: torch.randn(*s, device=dev, generator=g).to(dt).requires_grad_()
ins = dict(wa=r(B, D), wr=r(B, N, F, dt=bf), wt=torch.randn(B, N, F, device=dev), wm=(torch.rand(B, N, device=dev) < 0.75).float(),
           ws=r(B, P), wmu=r(B, S), wls=r(B, S), ra=r(B, D), rr=r(B, G), rt=torch.randn(B, G, device=dev),
           rm=(torch.rand(B, G, device=dev) < 0.75).float(), rs=r(B, P), rmu=r(B, S), rls=r(B, S),
           ls=torch.tensor(2.659, device=dev, requires_grad=True))
loss = MIRRORLoss()


def body():
    out = loss(ins["wa"], ins["wr"], ins["wt"], ins["wm"], ins["ws"].softmax(-1), ins["wmu"], ins["wls"], ins["ra"], ins["rr"], ins["rt"],
               ins["rm"], ins["rs"].softmax(-1), ins["rmu"], ins["rls"], ins["ls"].exp())
    out[0].backward()


for _ in range(3):
    body()
torch.cuda.synchronize()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        body()
    for _ in range(5):
        gr.replay()
    s.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(50):
        gr.replay()
    e1.record(s)
    s.synchronize()
print(f"loss fwd + bwd graph replay: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
