#!/usr/bin/env python3
"""Timeline view of a rocprofv3 *_kernel_trace.csv: per training step (delimited by adam_kernel) the wall time,
idle time and, per kernel category, the EXCLUSIVE time (nothing else running) and shared time."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    short = name.split("(")[0].replace("void ", "").split("<")[0]
    if "gemm_kernel" in name:
        gz = int(r["Grid_Size_Z"])
        short = "gemm(batched)" if gz >= 64 else "gemm"
    elif "at::native" in name or "rocclr" in name:
        short = "torch"
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short))
ev.sort()
# steps: split after each adam_kernel
steps, cur = [], []
for e in ev:
    cur.append(e)
    if e[2] == "adam_kernel":
        steps.append(cur)
        cur = []
steps = steps[2:]   # skip warm-up / profiled steps
excl, shared, walls, idles = collections.Counter(), collections.Counter(), [], []
for st in steps:
    pts = []
    for s, e, n in st:
        pts.append((s, 1, n))
        pts.append((e, -1, n))
    pts.sort()
    active = collections.Counter()
    t_prev = pts[0][0]
    idle = 0
    for t, d, n in pts:
        dt = t - t_prev
        if dt > 0:
            k = [a for a, c in active.items() if c > 0]
            if not k:
                idle += dt
            elif len(k) == 1 and active[k[0]] == 1:
                excl[k[0]] += dt
            else:
                for a in k:
                    shared[a] += dt / len(k)
        active[n] += d
        t_prev = t
    walls.append(pts[-1][0] - pts[0][0])
    idles.append(idle)
n = len(steps)
print(f"{n} steps: wall {sum(walls) / n / 1e6:.2f} ms/step, GPU idle {sum(idles) / n / 1e6:.2f} ms/step")
print("  exclusive  shared(split)  kernel")
for k in sorted(set(excl) | set(shared), key=lambda k: -(excl[k] + shared[k]))[:32]:
    print(f"{excl[k] / n / 1e6:10.3f} {shared[k] / n / 1e6:10.3f}   {k}")
