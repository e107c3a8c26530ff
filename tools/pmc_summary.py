#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (two separate runs, *_counter_collection.csv) -> per-kernel HBM-side
traffic per launch, corrected as MI355X_MICROARCH.md prescribes for gfx950: both counters are in KiB; FETCH_SIZE counts
128-byte fabric requests as 64 bytes for wide coalesced reads, so it is doubled; WRITE_SIZE is exact for 16-byte
streaming stores and float atomics.  Output: JSON {kernel: {fetch_kib_raw, write_kib, bytes_per_launch, launches}}."""
import collections
import csv
import json
import re
import sys


def norm(name: str) -> str:
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0]
    return re.sub(r"\s+", "", n.replace("unsigned short", "bf16"))


def load(fn):
    """Per-kernel totals, and the run cut into steps: a step ends with its adam_kernel dispatch (dispatch order)."""
    tot, cnt = collections.Counter(), collections.Counter()
    rows = list(csv.DictReader(open(fn)))
    if rows and "Dispatch_Id" in rows[0]:
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    steps, cur_n, cur_v = [], 0, 0.0
    for r in rows:
        k = norm(r["Kernel_Name"])
        v = float(r["Counter_Value"])
        tot[k] += v
        cnt[k] += 1
        cur_n += 1
        cur_v += v
        if k == "adam_kernel":
            steps.append((cur_n, cur_v))
            cur_n, cur_v = 0, 0.0
    return tot, cnt, steps


ft, fc, fsteps = load(sys.argv[1])
wt, wc, wsteps = load(sys.argv[2])
out = {}
for k in sorted(set(ft) | set(wt)):
    f = ft[k] / max(fc[k], 1)
    w = wt[k] / max(wc[k], 1)
    out[k] = {"fetch_kib_raw": round(f, 1), "write_kib": round(w, 1), "bytes_per_launch": int((2 * f + w) * 1024),
              "launches": int(max(fc[k], wc[k]))}
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_digest  # noqa: E402

# The run is 2 eager steps, the capture (dispatches nothing), 1 + 3 replays of the captured step, then 3 eager re-runs for the roofline
# leg.  A replayed step is what the bench times: its launches and bytes are reported on their own (the first step also pays one-time
# work, the eager ones run a few launches the capture does not hold, and the graph's instantiation uploads ~250 small copies once).
per_step = []
for (n1, f), (n2, w) in zip(fsteps, wsteps):
    per_step.append({"launches": n1, "hbm_bytes": int((2 * f + w) * 1024)})
replayed = None
if per_step:
    # the replays are the steps that repeat exactly (the most common launch count; eager steps differ from them by the two
    # Philox-state fills torch's graph replay adds and by the launches only an eager step runs)
    counts = collections.Counter(p["launches"] for p in per_step)
    lo = min(n for n, c in counts.items() if c == max(counts.values()))
    rp = [p for p in per_step if p["launches"] == lo]
    replayed = {"launches": lo, "hbm_bytes": int(sum(p["hbm_bytes"] for p in rp) / len(rp)), "steps_averaged": len(rp)}

json.dump({"csrc_sha256": csrc_digest(), "config": "c2", "steps": 9, "per_step": per_step, "replayed_step": replayed, "note": "bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB, averaged over the launches of each kernel in a "
                   "`bench.py --steps 3 --warmup 1` run = 9 steps (3 untimed, 3 timed, 3 eager re-run); see tools/pmc_summary.py", "kernels": out}, sys.stdout, indent=1)
