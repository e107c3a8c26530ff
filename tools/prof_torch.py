#!/usr/bin/env python3
"""torch-native kernels of one training step from a rocprofv3 kernel trace: longest launches and totals per kernel."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
step = rows[ends[-2] + 1: ends[-1] + 1]
tot = collections.Counter()
out = []
for r in step:
    n = r["Kernel_Name"]
    if "at::native" not in n and "rocclr" not in n:
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    short = n.split("<")[1].split(">")[0][:60] if "Functor" in n or "kernel" in n else n[:60]
    key = (n.split("(")[0][-70:], r["Stream_Id"])
    tot[key] += d
    out.append((d, r["Stream_Id"], int(r["Grid_Size_X"]), n[:110]))
for d, st, g, n in sorted(out, reverse=True)[:25]:
    print(f"{d:8.1f} us s{st} grid={g:9d} {n}")
print("stream totals:", {s: round(sum(v for (k, ss), v in tot.items() if ss == s), 1) for s in {k[1] for k in tot}})
