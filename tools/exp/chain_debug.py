#!/usr/bin/env python3
"""Debug aid: run the chain forward + backward on fixed inputs and dump the work buffers (compare MH_CHAIN_Q=0 vs 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mirror_amd import kernels as K
torch.manual_seed(3)
BH, m, iters = 2, 256, 6
x = (torch.randn(1, BH, m, m) * 2).softmax(-1).cuda()
st = K.pinv_absmax(x)
saved = torch.zeros((iters, 4, BH, m, m), device="cuda", dtype=torch.bfloat16)
z0, xb = K.pinv_chain_prep(x, st, saved[0, 0])
zfT = torch.empty((BH, m, m), device="cuda", dtype=torch.bfloat16)
K.pinv_chain_fwd(xb, saved, zfT, iters)
G = torch.randn(BH, m, m).cuda()
work = torch.zeros_like(saved)
dX = torch.empty((BH, m, m), device="cuda"); dz0 = torch.empty((BH, m, m), device="cuda")
K.pinv_chain_bwd(xb, saved, K.pinv_chain_pack(G), work, dX, dz0, iters)
torch.cuda.synchronize()
torch.save({"G": G.cpu(), "saved": saved.cpu(), "work": work.cpu(), "dX": dX.cpu(), "dz0": dz0.cpu(), "zfT": zfT.cpu()}, sys.argv[1])
