R=$PWD; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INSTS_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA"; do
  i=$((i+1)); rm -rf /tmp/pn$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/pn$i -o p -- python3 $R/tools/exp/gemm_n96.py n128 > /tmp/pn$i.log 2>&1
  f=$(find /tmp/pn$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_kernel" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in acc.items():
    print({c: f"{sum(v)/len(v):.4g}" for c, v in cs.items()})
PY
  else echo "set $i failed: $(tail -2 /tmp/pn$i.log)"; fi
done
