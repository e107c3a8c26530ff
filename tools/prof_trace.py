#!/usr/bin/env python3
"""Categorise a rocprofv3 *_kernel_trace.csv per training step (pinv-sized GEMMs are recognised by their grid)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
cat_t, cat_n = collections.Counter(), collections.Counter()
for r in rows:
    name = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    gx = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
    gy, gz = int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])
    blocks = gx * gy * gz
    if "gemm_kernel" in name:
        if gz >= 64 and gx == 4:
            c = "gemm: pinv 256^3 batched"
        elif gz >= 64:
            c = "gemm: attention batched (sims, a.v, ...)"
        elif blocks <= 256:
            c = "gemm: small grid (<=256 blocks)"
        else:
            c = "gemm: projections / wgrads"
    elif "skinny" in name or "transpose" in name:
        c = "skinny linears"
    elif "at::native" in name or "rocclr" in name:
        c = "torch glue (add/fill/copy)"
    else:
        c = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").split("<")[0]
    cat_t[c] += d
    cat_n[c] += 1
tot = sum(cat_t.values())
print(f"total {tot / steps:.2f} ms/step, {len(rows) / steps:.0f} launches/step")
for c, t in cat_t.most_common(28):
    print(f"{t / steps:7.3f} ms/step {cat_n[c] / steps:7.1f} calls  {c}")
