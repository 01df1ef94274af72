"""MFMA utilisation per kernel from one rocprofv3 PMC pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES).

usage: python scripts/pmc_mfma.py <counter_collection.csv>
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256 CUs * 4 SIMDs)   (the gfx94x derived-counter formula the
guide says gfx950 falls back to, MI355X_MICROARCH.md 'rocprofv3 PMC slots')."""
import csv
import json
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for r in rows:
    n = r["Kernel_Name"]
    if "gemm_rows_kernel<" in n:
        k = "gemm_rows<" + n.split("gemm_rows_kernel<")[1].split(">")[0] + ">"
    elif "attn_rows_kernel" in n:
        k = "attn_rows"
    elif "rmsnorm_rows" in n:
        k = "rmsnorm_rows"
    elif "sampler" in n:
        k = "sampler"
    else:
        k = n.split("(")[0][-40:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r["Dispatch_Id"])
out = {}
for k, c in acc.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    out[k] = {"dispatches": len(calls[k]), "GRBM_GUI_ACTIVE": int(gui), "SQ_VALU_MFMA_BUSY_CYCLES": int(busy),
              "SQ_INSTS_VALU_MFMA_MOPS_BF16": int(c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)),
              "SQ_BUSY_CU_CYCLES": int(c.get("SQ_BUSY_CU_CYCLES", 0.0)),
              "MfmaUtil_percent": round(100.0 * busy / (gui * 256 * 4), 3) if gui else None}
print(json.dumps(out, indent=1))
