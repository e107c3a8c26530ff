#!/usr/bin/env python3
"""torch-native launches of the last step of an eager-mode rocprofv3 kernel trace, grouped by (functor, size), per stream."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
step = rows[ends[-2] + 1: ends[-1] + 1]
main = step[-1]["Stream_Id"]
c, t = collections.Counter(), collections.Counter()
for r in step:
    n = r["Kernel_Name"]
    if "at::native" in n or "rocclr" in n:
        key = ("main" if r["Stream_Id"] == main else "side", n.split("<")[1].split(">")[0][:60] if "<" in n else n[:60], r["Grid_Size_X"])
        c[key] += 1
        t[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = collections.Counter()
for k, v in t.items():
    tot[k[0]] += v
print({k: round(v, 1) for k, v in tot.items()}, "us;", sum(c.values()), "launches")
for k, v in sorted(t.items(), key=lambda kv: -kv[1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{v:8.1f} us  x{c[k]:3d}  {k[0]:4s} threads={k[2]:>8s}  {k[1]}")
