#!/usr/bin/env python3
"""One fused RNA Block (B = 16, D = 512, Hh = 2048), forward + backward, 30 times: run under
`rocprofv3 --kernel-trace --output-format csv` and feed the trace to this script with --report to list the launches of the
last iteration in order (name, grid, LDS, duration)."""
import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    rows = [r for r in rows if "rna_" in r["Kernel_Name"] or "headattn" in r["Kernel_Name"]]
    per = 11
    last = rows[-per:]
    t0 = int(last[0]["Start_Timestamp"])
    for r in last:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:6.1f} us  grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):4d}  lds {r['LDS_Block_Size']:>6}  {n}")
    sys.exit(0)
import importlib
import torch
from mirror_amd import functional as Fn
mm = importlib.import_module("mirror_amd.models.mirror")
torch.manual_seed(0)
blk = mm.Block(512, 8, 4.0, True, 0.1, 1e-6).cuda().train()
x = torch.randn(16, 512, device="cuda", requires_grad=True)
dy = torch.randn(16, 512, device="cuda")
prec = Fn.POLICIES["bf16"]
for _ in range(30):
    y = blk(x, prec)
    y.backward(dy)
    torch.cuda.synchronize()
