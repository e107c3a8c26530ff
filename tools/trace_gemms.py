#!/usr/bin/env python3
"""Print every K.gemm call of one eager training step at a bench workload: shapes, strides, variant, time (HIP events, the call
alone on the chip).  usage: python3 tools/trace_gemms.py [c2|c1|template]"""
import os, sys, collections
os.environ["MIRROR_GRAPH"] = "0"
os.environ["MIRROR_RNA_GRAPH"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mirror_amd import kernels as K
from bench import CONFIGS

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
calls = []
orig = K.gemm
def spy(a, b, *args, **kw):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(a, b, *args, **kw)
    e1.record()
    torch.cuda.synchronize()
    var = "R" * bool(kw.get("R") is not None) + "b" * bool(kw.get("bias") is not None) + (f" alpha={kw['alpha']}" if kw.get("alpha", 1.0) != 1.0 else "")
    calls.append((tuple(a.shape), tuple(a.stride()), tuple(b.shape), tuple(b.stride()), bool(kw.get("accumulate", False)),
                  str(r.dtype).replace("torch.", ""), var, e0.elapsed_time(e1) * 1e3))
    return r
K.gemm = spy
import mirror_amd.functional as Fn
Fn.K.gemm = spy
import mirror_amd.models as M
from mirror_amd.engine import TrainEngine
from mirror_amd.losses import MIRRORLoss
dev = torch.device("cuda", 0)
torch.manual_seed(42)
shp = CONFIGS[cfg]
model = M.mirror(wsi_embed_dim=shp["F"], rna_embed_dim=shp["G"], embed_dim=shp["D"], wsi_num_tokens=shp["N"],
                 rna_encoder_depth=shp["L"], rna_mlp_ratio=shp["mlp"], rna_norm_layer="layernorm", rna_act_layer="gelu",
                 rna_num_heads=shp["heads"]).to(dev).train()
loss_fn = MIRRORLoss(alignment_loss_weight=0.5, wsi_retention_loss_weight=0.15, rna_retention_loss_weight=0.15,
                     style_loss_weight=0.1, cluster_loss_weight=0.1, gather_distributed=False)
eng = TrainEngine(model, loss_fn, lr=2e-5, precision="bf16", graph=False)
wsi = torch.randn(16, shp["N"], shp["F"], device=dev).bfloat16()
rna = torch.randn(16, shp["G"], device=dev)
for _ in range(2):
    eng.step(wsi, rna)
calls.clear()
eng.step(wsi, rna)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for c in calls:
    k = c[:7]
    n, t = agg.get(k, (0, 0.0))
    agg[k] = (n + 1, t + c[7])
tot = sum(t for _, t in agg.values())
print(f"{len(calls)} gemm calls, {tot / 1e3:.2f} ms in all (each timed alone)")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    a, sa, b, sb, acc, dt, var = k
    bt = 1
    for x in a[:-2]:
        bt *= x
    gf = 2.0 * bt * a[-2] * a[-1] * b[-1] / 1e9
    print(f"{t:9.1f} us {n:3d}x {t / n:8.1f} us {gf / (t / n) * 1e3 if t else 0:7.1f} TF/s  A{a}{sa} B{b}{sb} acc={int(acc)} -> {dt}  {var}")
