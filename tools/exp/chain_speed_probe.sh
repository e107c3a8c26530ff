#!/bin/bash
# What would a faster Moore-Penrose chain buy INSIDE the step?  EXP build (never shipped): the chain kernels run fewer iterations
# (results garbage), everything else unchanged.  bash tools/exp/chain_speed_probe.sh   (GPU box, repo root; rebuilds the library twice)
R=$PWD; make -C mirror_amd/csrc -B EXP=1 -j16 > /dev/null 2>&1 || exit 1
cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
run() { echo -n "$1: "; env $1 python3 $R/tools/exp/step_time.py 30 2>/dev/null | tail -1; }
for i in 1 2; do
run X=0
run MH_EXP_CHAIN_BWD_ITERS=5
run MH_EXP_CHAIN_BWD_ITERS=4
run MH_EXP_CHAIN_FWD_ITERS=4
run "MH_EXP_CHAIN_BWD_ITERS=4 MH_EXP_CHAIN_FWD_ITERS=4"
done
cd $R; make -C mirror_amd/csrc -B -j16 > /dev/null 2>&1
