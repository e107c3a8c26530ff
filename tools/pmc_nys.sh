#!/bin/bash
# PMC passes over tools/bench_nys.py: where do the fused-attention kernels spend their cycles?
# usage (on the GPU box): bash tools/pmc_nys.sh  -> prints per-kernel counter averages
R=$PWD; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY"; do
  rm -rf /tmp/pmc; rocprofv3 --pmc $set --output-format csv -d /tmp/pmc -o p -- python3 $R/tools/bench_nys.py > /dev/null 2>&1
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    if "nys_" not in n: continue
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in sorted(acc.items()):
    print(n, {c: f"{sum(v)/len(v):.3g}" for c, v in cs.items()})
PY
done
