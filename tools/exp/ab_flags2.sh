# second set: the chain-branch placement hooks, one at a time against the default, alternating (tools/exp/flag_time.py)
export PYTHONPATH=$PWD
run() { printf '%-44s %s ms\n' "${1:-default}" "$(python3 tools/exp/flag_time.py 30 $1 2>/dev/null | tail -1)"; }
for rep in 1 2; do
run ""
for f in functional._SIM2_SIDE=False functional._S2_SIDE=False functional._LM_MERGE_LATE=False functional._W2_ON_CHAIN=False functional._DEFER_V=False models.mirror._RNA_LATE=False; do run $f; done
done
run ""
