#!/usr/bin/env python3
"""Summarise a rocprofv3 *_kernel_stats.csv per training step: python tools/prof_summary.py <csv> <steps_in_run> [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / steps / 1e6:.3f} ms/step, {sum(int(r['Calls']) for r in rows) / steps:.0f} launches/step")
for r in rows[:top]:
    t, c = int(r["TotalDurationNs"]), int(r["Calls"])
    print(f"{t / steps / 1e6:7.3f} ms/step {c / steps:6.1f} calls avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:118]}")
