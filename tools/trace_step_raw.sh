#!/bin/bash
# Raw rocprofv3 kernel trace of a few graph-replayed bench steps, copied to gpurun_out/<tag>_kernel_trace.csv for offline analysis
# (tools/prof_step_listing.py).  usage (GPU box, repo root): bash tools/trace_step_raw.sh <tag> [extra bench args]
TAG=${1:-r02_x}; shift; R=$PWD; mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/p3; rocprofv3 --kernel-trace --output-format csv -d /tmp/p3 -o r -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_bench_under_trace.json 2>/dev/null
t=$(find /tmp/p3 -name "*kernel_trace.csv" | head -1)
cp $t $R/gpurun_out/${TAG}_kernel_trace.csv
wc -l $R/gpurun_out/${TAG}_kernel_trace.csv; cat $R/gpurun_out/${TAG}_bench_under_trace.json | cut -c1-200
