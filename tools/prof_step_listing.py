#!/usr/bin/env python3
"""List the kernels of ONE step of a rocprofv3 kernel trace in start order: start offset, duration, what else was running.
usage: prof_step_listing.py <kernel_trace.csv> [step_from_end=1 | fastest]
`fastest`: the step with the shortest wall span — in a bench trace a graph-REPLAYED step (the eager re-runs of the roofline leg
at the end of the run, and the warm-up steps at its start, are longer and hold a few launches the captured step does not)."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
sel = sys.argv[2] if len(sys.argv) > 2 else "1"
back = 1 if sel == "fastest" else int(sel)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step ends with the (last) adam_kernel launch
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
# group consecutive adam launches
marks = [i for k, i in enumerate(ends) if k + 1 == len(ends) or ends[k + 1] - i > 50]
if sel == "fastest":
    spans = []
    for k in range(1, len(marks)):
        seg = rows[marks[k - 1] + 1:marks[k] + 1]
        spans.append((max(int(r["End_Timestamp"]) for r in seg) - int(seg[0]["Start_Timestamp"]), k))
    back = len(marks) - min(spans)[1]
hi = marks[-back]
lo = marks[-back - 1] + 1
step = rows[lo:hi + 1]
t0 = int(step[0]["Start_Timestamp"])


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_:]+)(<[^(]*>)?\(", n)
    base = m.group(1) if m else n.split("(")[0]
    tpl = (m.group(2) or "") if m else ""
    tpl = tpl.replace("unsigned short", "bf").replace("float", "f").replace("true", "1").replace("false", "0").replace(" ", "")
    return (base.split("::")[-1] + tpl)[:58]


ivs = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0) for r in step]
print(f"step: {len(step)} launches, {(max(e for _, e in ivs)) / 1e3:.1f} us")
for r, (s, e) in zip(step, ivs):
    others = [short(q["Kernel_Name"]).split("<")[0] for q, (s2, e2) in zip(step, ivs) if q is not r and s2 < e and e2 > s and min(e, e2) - max(s, s2) > 0.3 * (e - s)]
    gx = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
    grid = f"{gx}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}"
    print(f"{s / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{r.get('Queue_Id', '?')} {short(r['Kernel_Name']):58s} {grid:14s} | {','.join(sorted(set(others)))[:80]}")
