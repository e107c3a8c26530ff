#!/bin/bash
# MFMA-busy share per kernel over a 3-step bench run: SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES (own PMC pass, no trace
# domains), keyed by the digest of the kernel sources.  usage (GPU box, repo root): bash tools/pmc_mfma.sh > gpurun_out/pmc_mfma_busy.json
# (copy to profiles/pmc_mfma_busy.json when it is the evidence for HEAD: bench.py reads it next to profiles/pmc_traffic.json)
R=$PWD; export MIRROR_ROOT=$R; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pm; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -o m -- python3 $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
out = {}
for k, c in acc.items():
    busy, mfma = c.get("SQ_BUSY_CU_CYCLES", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if mfma <= 0: continue
    out[k] = {"launches": n[k], "mfma_busy_cycles": mfma, "cu_busy_cycles": busy,
              "bf16_mfma_mops": c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0), "gui_active_cycles": c.get("GRBM_GUI_ACTIVE", 0.0)}
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    # MFMA pipe utilisation of the whole chip while the kernel ran: busy MFMA cycles (32 per 32x32x16 bf16 instruction, summed
    # over SIMDs) / (kernel cycles x 1024 SIMDs); GRBM_GUI_ACTIVE is reported summed over the 8 XCDs
    out[k]["mfma_pipe_utilisation_of_chip"] = round(mfma / (gui / 8.0 * 1024.0), 4) if gui else None
import os
sys.path.insert(0, os.environ["MIRROR_ROOT"])
from bench import csrc_digest
print(json.dumps({"csrc_sha256": csrc_digest(), "config": "c2",
                  "note": "sums over all launches of a 3+3-step bench run (eager warm steps + graph replays); SQ counters are summed over XCDs / SEs as rocprofv3 reports them; "
                          "mfma_pipe_utilisation_of_chip = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)",
                  "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles"]))}, indent=1))
PY
