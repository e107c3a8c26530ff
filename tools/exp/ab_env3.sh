R=$PWD; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
run() { echo -n "$1: "; env $1 python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'; }
for i in 1 2 3; do for v in "$@"; do run $v; done; done
