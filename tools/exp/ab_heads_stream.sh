set -e
for v in 0 1 0 1; do
  echo "== MIRROR_HEADS_SIDE=$v"
  MIRROR_HEADS_SIDE=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/ab_err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
echo "== eager"; MIRROR_GRAPH=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/ab_err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
python -m pytest tests/test_model_gpu.py tests/test_engine_gpu.py tests/test_bench_path_gpu.py -q -m gpu -x --timeout 900 2>&1 | tail -5
