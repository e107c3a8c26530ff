# grid of the LayerNorm backward (partial gamma / beta sums per block, HBM-bound): MH_LN_BWD_BLOCKS sweep, same box
for v in 512 1024 256 2048 512 1024; do echo -n "LN_BWD_BLOCKS=$v  "; MH_LN_BWD_BLOCKS=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
