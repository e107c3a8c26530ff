#!/bin/bash
# HBM-side traffic per kernel and per step (run on the GPU box from the repo root): two PMC passes (FETCH_SIZE, WRITE_SIZE in
# separate runs: they do not fit one pass, MI355X_MICROARCH.md rocprofv3 PMC slots) of `bench.py --steps 3 --warmup 1`
# -> gpurun_out/pmc_traffic.json (copy to profiles/ when it is the evidence for HEAD) + a per-step table.  usage: bash tools/pmc_traffic.sh [bench args]
set -u
R=$PWD; mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
rm -rf /tmp/p3 /tmp/p4
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p3 -o f -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p4 -o w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
ff=$(find /tmp/p3 -name "*counter_collection.csv" | head -1); fw=$(find /tmp/p4 -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python3 $R/tools/pmc_summary.py $ff $fw > $R/gpurun_out/pmc_traffic.json
python3 - $R/gpurun_out/pmc_traffic.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
steps = 9.0      # 3 untimed (>= 3 always) + 3 timed + 3 eager re-run steps of `bench.py --steps 3 --warmup 1`
rows = sorted(((v["bytes_per_launch"] * v["launches"], v["bytes_per_launch"], v["launches"], k) for k, v in d["kernels"].items()), reverse=True)
tot = sum(r[0] for r in rows)
print(f"csrc {d['csrc_sha256']}: {tot / steps / 1e9:.2f} GB per step over {steps:.0f} steps, {sum(r[2] for r in rows) / steps:.0f} launches per step")
for r in rows[:40]:
    print(f"{r[0] / steps / 1e9:7.3f} GB/step {r[1] / 1e6:8.1f} MB x {r[2] / steps:5.1f}  {r[3][:100]}")
PY
