# HIP_FORCE_DEV_KERNARG unset / 0 / 1, same box (kernel arguments in device memory: shorter dispatch of dependent launches)
run() { python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do
  echo -n "unset  "; (unset HIP_FORCE_DEV_KERNARG; run)
  echo -n "=1     "; HIP_FORCE_DEV_KERNARG=1 run
  echo -n "=0     "; HIP_FORCE_DEV_KERNARG=0 run
done
env | grep -i "^HIP_\|^HSA_\|^AMD_\|^GPU_\|^ROC" 
