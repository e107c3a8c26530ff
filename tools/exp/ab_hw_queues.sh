# GPU_MAX_HW_QUEUES (default 4): how the step's streams / graph branches map onto hardware queues, same box
run() { python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || echo failed; }
for q in 4 3 5 6 1 4; do echo -n "GPU_MAX_HW_QUEUES=$q  graph: "; GPU_MAX_HW_QUEUES=$q run; done
for q in 4 3 5 8; do echo -n "GPU_MAX_HW_QUEUES=$q  eager: "; MIRROR_GRAPH=0 GPU_MAX_HW_QUEUES=$q run; done
