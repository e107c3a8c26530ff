for b in tools/exp/chain_exp_*.bin; do timeout -k 5 60 $b || echo "$b failed"; done
