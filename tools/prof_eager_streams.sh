#!/bin/bash
# Eager-launch kernel trace of a few bench steps split by HIP stream (RNA side stream, chain side stream, main stream).
# usage (GPU box, repo root): bash tools/prof_eager_streams.sh <tag>     -> gpurun_out/<tag>_per_stream.txt, _timeline.txt
TAG=${1:-r02_x}; R=$PWD; mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/p2; MIRROR_GRAPH=0 MIRROR_RNA_GRAPH=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/p2 -o r -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
t=$(find /tmp/p2 -name "*kernel_trace.csv" | head -1)
python3 $R/tools/prof_streams.py $t 16 > $R/gpurun_out/${TAG}_per_stream.txt 2>&1
python3 $R/tools/prof_timeline.py $t > $R/gpurun_out/${TAG}_timeline.txt 2>&1
cat $R/gpurun_out/${TAG}_per_stream.txt
