#!/bin/bash
# A/B of HIP-runtime graph switches on the replayed c2 step (same box): bash tools/exp/ab_graph_knobs.sh
R=$PWD; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
run() { echo -n "$1: "; env $1 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'; }
run X=0
for v in "$@"; do run $v; done
run X=0
