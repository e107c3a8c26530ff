#!/bin/bash
# DOT dump of the captured whole-step HIP graph (runtime switch DEBUG_HIP_GRAPH_DOT_PRINT): bash tools/exp/graph_dot.sh <tag>
TAG=${1:-r03}; R=$PWD; mkdir -p $R/gpurun_out; rm -rf /tmp/dot; mkdir -p /tmp/dot; cd /tmp/dot; export TMPDIR=/tmp PYTHONPATH=$R
DEBUG_HIP_GRAPH_DOT_PRINT=1 python3 $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline > bench.json 2> err.txt
ls -la /tmp/dot | head -20
for f in /tmp/dot/graph_*; do n=$(grep -c -- "->" $f); echo "$f edges=$n"; done
big=$(ls -S /tmp/dot/graph_* 2>/dev/null | head -1)
[ -n "$big" ] && cp $big $R/gpurun_out/${TAG}_step_graph.dot && wc -c $R/gpurun_out/${TAG}_step_graph.dot
