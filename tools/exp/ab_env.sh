# usage: bash tools/exp/ab_env.sh VAR [bench args]: alternates VAR=0 / VAR=1 twice in one box
V=$1; shift
for v in 0 1 0 1; do
  echo -n "$V=$v  "
  env $V=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
