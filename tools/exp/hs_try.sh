for v in "MIRROR_RNA_LATE=0" "MIRROR_RNA_LATE=1" "MIRROR_EXP_NO_SIDE=1" ; do
  echo "== $v"
  env $v timeout -k 10 200 python3 -X faulthandler bench.py --config c1 --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/hs_out.txt 2> gpurun_out/hs_err.txt; echo rc=$?
  grep -A6 "Current thread" gpurun_out/hs_err.txt | head -8; cut -c1-120 gpurun_out/hs_out.txt
done
