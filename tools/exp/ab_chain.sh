#!/bin/bash
# the chain backward variants alone on the chip, interleaved on one box: bash tools/exp/ab_chain.sh
export PYTHONPATH=$PWD
for i in 1 2 3; do for v in 0 1 2; do echo -n "MH_CHAIN_BWD2=$v: "; MH_CHAIN_BWD2=$v python3 tools/bench_chain.py 2>&1 | grep "^pinv_chain_bwd"; done; done
