#!/bin/bash
# N=1 step time with the HIP graph on and off (the off figure is what an eager multi-GPU rank pays in launches)
for g in 1 0; do
  MIRROR_GRAPH=$g python bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph=$g', d['value'], d['ms_per_step'])"
done
