set -e
for late in 0 1 0 1; do
  for g in 1 0; do
    echo "== late=$late graph=$g"
    MIRROR_BENCH_HOSTTIME=1 MIRROR_RNA_LATE=$late MIRROR_GRAPH=$g python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/ab_err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
    grep "host us" gpurun_out/ab_err.txt | cut -c1-200
  done
done
