#!/usr/bin/env python3
"""Longest individual launches of one training step (the step before the last adam_kernel) from a kernel trace."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]
out = []
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0][:70]
    wg = int(r["Workgroup_Size_X"])
    grid = (int(r["Grid_Size_X"]) // max(wg, 1), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    out.append((d, name, grid, wg, r["Stream_Id"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for d, name, grid, wg, st in sorted(out, reverse=True)[:n]:
    print(f"{d:9.1f} us  s{st}  {name:70s} grid={grid} wg={wg}")
