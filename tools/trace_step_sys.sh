#!/bin/bash
# kernel + memory-copy + HIP runtime API trace of a few bench steps (no counters), CSVs copied to gpurun_out/<tag>_*.csv
# usage (GPU box, repo root): bash tools/trace_step_sys.sh <tag> [extra bench args]
TAG=${1:-r02_x}; shift; R=$PWD; mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/p4; rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d /tmp/p4 -o r -- python3 $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_bench_under_trace.json 2>/dev/null
for k in kernel_trace memory_copy_trace hip_api_trace; do
  t=$(find /tmp/p4 -name "*${k}.csv" | head -1); [ -n "$t" ] && cp $t $R/gpurun_out/${TAG}_${k}.csv
done
ls -la $R/gpurun_out/${TAG}_*; cut -c1-200 $R/gpurun_out/${TAG}_bench_under_trace.json
