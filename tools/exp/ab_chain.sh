#!/bin/bash
# the chain variants alone on the chip, interleaved on one box: bash tools/exp/ab_chain.sh
export PYTHONPATH=$PWD
for i in 1 2; do
  echo -n "old bwd   : "; MH_CHAIN_BWD2=0 python3 tools/bench_chain.py 2>&1 | grep "^pinv_chain_fwd\|^pinv_chain_bwd" | tr '\n' ' '; echo
  echo -n "bwd2      : "; python3 tools/bench_chain.py 2>&1 | grep "^pinv_chain_fwd\|^pinv_chain_bwd" | tr '\n' ' '; echo
  echo -n "quarter   : "; MH_CHAIN_Q=1 python3 tools/bench_chain.py 2>&1 | grep "^pinv_chain_fwd\|^pinv_chain_bwd" | tr '\n' ' '; echo
done
