#!/bin/bash
# usage: bash tools/prof_kernels.sh <tag> [bench args...]: rocprofv3 kernel stats of a short bench run -> gpurun_out/<tag>_kernel_stats.csv
# (+ a top-40 table by total time with per-step launch counts and microseconds per step)
set -u
TAG=$1; shift
R=$PWD; mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
rm -rf /tmp/pk_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk_$TAG -o r -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2>/dev/null
f=$(find /tmp/pk_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/${TAG}_kernel_stats.csv
python3 - "$R/gpurun_out/${TAG}_kernel_stats.csv" 16 <<'PY' > $R/gpurun_out/${TAG}_kernel_top.txt
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])      # 3 untimed + 10 timed + min(10, 5) eager re-run... = bench.py's step count under these flags
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over ~{steps:.0f} steps = {tot/1e6/steps:.3f} ms/step")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:60]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"\(.*", "", n)[:78]
    print(f"{float(r['TotalDurationNs'])/1e3/steps:9.1f} us/step {int(r['Calls'])/steps:6.1f} x {float(r['AverageNs'])/1e3:8.1f} us  {n}")
PY
cat $R/gpurun_out/${TAG}_kernel_top.txt | head -70
