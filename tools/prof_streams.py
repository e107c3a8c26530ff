#!/usr/bin/env python3
"""Per-stream kernel time of the last complete training step in a rocprofv3 kernel trace (eager launch mode, where the
stream ids survive): sum of durations, launches, and the top kernels of each stream."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
step = rows[ends[-2] + 1: ends[-1] + 1]
wall = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6
print(f"step wall {wall:.2f} ms, {len(step)} launches")
per = collections.defaultdict(lambda: [0.0, 0, collections.Counter()])
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").split("<")[0]
    if "at::native" in r["Kernel_Name"]:
        n = "torch"
    p = per[r["Stream_Id"]]
    p[0] += d; p[1] += 1; p[2][n] += d
for s, (t, n, c) in sorted(per.items()):
    print(f"stream {s}: {t:.2f} ms in {n} launches")
    for k, v in c.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
        print(f"      {v:7.3f}  {k}")
