# the step with one test hook flipped at a time against the default, alternating (tools/exp/flag_time.py)
export PYTHONPATH=$PWD
run() { printf '%-44s %s ms\n' "${1:-default}" "$(python3 tools/exp/flag_time.py 30 $1 2>/dev/null | tail -1)"; }
for rep in 1 2; do
run ""
for f in functional._TAIL_ASIDE=False engine._DEFER_SKINNY=False engine._TRANSPOSE_AT_START=False models.mirror._HEADS_SIDE=False; do run $f; done
done
run ""
