R=$PWD; cd /tmp; export TMPDIR=/tmp
python3 $R/tools/exp/gemm_n96.py n96; python3 $R/tools/exp/gemm_n96.py n128
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1)); rm -rf /tmp/pm$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/pm$i -o p -- python3 $R/tools/exp/gemm_n96.py n96 > /dev/null 2>&1
  f=$(find /tmp/pm$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_kernel" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in acc.items():
    print(n, {c: f"{sum(v)/len(v):.4g}" for c, v in cs.items()})
PY
done
