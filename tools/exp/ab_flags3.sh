# usage: bash tools/exp/ab_flags3.sh "flagA" "flagB" ... : default and each flag setting alternated, three rounds
export PYTHONPATH=$PWD
run() { printf '%-60s %s ms\n' "${1:-default}" "$(python3 tools/exp/flag_time.py 30 $1 2>/dev/null | tail -1)"; }
for rep in 1 2 3; do
run ""
for f in "$@"; do run "$f"; done
done
run ""
