# ROCm runtime switches that could change dispatch latency of the replayed step, same box
run() { python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || echo failed; }
echo -n "baseline                          "; run
echo -n "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0  "; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run
echo -n "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1  "; DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 run
echo -n "GPU_MAX_HW_QUEUES=8               "; GPU_MAX_HW_QUEUES=8 run
echo -n "GPU_MAX_HW_QUEUES=2               "; GPU_MAX_HW_QUEUES=2 run
echo -n "HSA_ENABLE_INTERRUPT=0            "; HSA_ENABLE_INTERRUPT=0 run
echo -n "AMD_SERIALIZE_KERNEL=0 HIP_LAUNCH_BLOCKING=0 "; run
echo -n "baseline                          "; run
