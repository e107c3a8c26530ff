# workgroups per (b, h) of the sequence-walking attention kernels (attn1 fwd, attn1 bwd dq, attn3 bwd dk/dv)
for v in "" 2 8 1 "" 2 16; do echo -n "WALKERS=$v  "; MH_NYS_WALKERS=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
