#!/bin/bash
# LDS bank conflicts of gemm_big_kernel with (gemm_e0) and without (gemm_e4) its epilogue: tools/exp/bin binaries of tools/exp/gemm_exp.cpp
R=$PWD; cd /tmp; export TMPDIR=/tmp
for b in gemm_e0 gemm_e4; do
  rm -rf /tmp/pmc; rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d /tmp/pmc -o p -- $R/tools/exp/bin/$b > /dev/null 2>&1
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  python3 - "$f" $b <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r["Grid_Size"], r["Kernel_Name"][:60])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in sorted(acc.items()):
    print(sys.argv[2], n, {c: f"{sum(v)/len(v):.4g}" for c, v in cs.items()})
PY
done
