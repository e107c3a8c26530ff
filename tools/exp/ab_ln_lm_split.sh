for r in 1 2 3 4 5 6; do
  if [ $((r % 2)) -eq 1 ]; then order="1 2"; else order="2 1"; fi
  for v in $order; do
    echo -n "round $r MH_LN_LM_SPLIT=$v  "
    MH_LN_LM_SPLIT=$v python3 bench.py --config c4 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
