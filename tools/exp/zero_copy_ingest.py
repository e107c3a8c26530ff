#!/usr/bin/env python3
"""Host -> HBM ingest by a KERNEL that reads pinned host memory through the GPU's mapping of it (no hipMemcpy): does it run at the
PCIe rate, and does it overlap compute on another stream where hipMemcpyAsync does not (tools/exp/h2d_overlap.py)?
Measured: 55 GB/s alone (correct bytes), but beside ten 8192^3 GEMMs on the main stream the pair takes 16-17 ms against 9.6 + 2.5
serial — also with a dedicated 8 ... 128-workgroup streaming kernel: host reads in flight slow the HBM-side traffic of the
compute kernels down on this platform, so the feed stays a hipMemcpyAsync in front of the step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mirror_amd import _lib
from mirror_amd._lib import MH_BF16, MH_F32
dev = torch.device("cuda", 0)
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
n = 16 * 4096 * 1024
h = torch.randn(n).to(torch.bfloat16).pin_memory()
d = torch.zeros(n, device=dev, dtype=torch.bfloat16)
side = torch.cuda.Stream(device=dev)
lib = _lib.load()
def ingest(stream):      # any elementwise kernel will do as the reader: mh_cast bf16 -> bf16 is a copy
    _lib.call("mh_cast", h.data_ptr(), d.data_ptr(), n, MH_BF16, MH_BF16, stream=stream.cuda_stream)
def work(k):
    for _ in range(k):
        torch.mm(a, a)
def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
work(3)
ingest(torch.cuda.current_stream()); torch.cuda.synchronize()
print("bytes equal:", bool(torch.equal(d.cpu(), h)))
tw = t(lambda: work(10))
ti = t(lambda: ingest(torch.cuda.current_stream()))
def both():
    side.wait_stream(torch.cuda.current_stream())
    ingest(side)
    work(10)
    torch.cuda.current_stream().wait_stream(side)
tb = t(both)
print(f"work {tw:.2f} ms, ingest kernel {ti:.2f} ms ({n * 2 / ti / 1e6:.1f} GB/s), both {tb:.2f} ms")
