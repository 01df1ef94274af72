"""MFMA utilisation per kernel from one rocprofv3 PMC pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES).

usage: python scripts/pmc_mfma.py <counter_collection.csv>

Units, calibrated on a kernel whose MFMA count is known (round 2: one prefill group's gate_up GEMM, 32768 rows x 8192 x
1024 = 33.55 M v_mfma_f32_16x16x32_bf16): SQ_VALU_MFMA_BUSY_CYCLES reads 536.9 M = 16 per MFMA, the clocks one such
MFMA occupies a SIMD's matrix core at the 2.5 PFLOP/s dense rating (8192 MACs / 509 MACs per clock and SIMD);
GRBM_GUI_ACTIVE is the SUM over the 8 XCDs (15-25 "GHz" against the kernel's wall time).  So
    MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * 256 CUs * 4 SIMDs),
8 x the gfx94x derived-counter formula (BUSY / (GUI * CUs * 4)) that round 1 printed.  Both are reported.  Cross-check on
the same kernel: 5.5e14 FLOP in 871 us (profiled) = 0.63 PFLOP/s; GUI / 8 = 1.67 M cycles = 1.92 GHz, at which the
matrix cores peak at 2.0 PFLOP/s: 31 %, the figure the formula gives."""
import csv
import json
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for r in rows:
    n = r["Kernel_Name"]
    if "gemm_as_kernel<" in n:
        k = "gemm_as<" + n.split("gemm_as_kernel<")[1].split(">")[0] + ">"
    elif "gemm_wide_kernel<" in n:
        k = "gemm_wide<" + n.split("gemm_wide_kernel<")[1].split(">")[0] + ">"
    elif "gemm_rows_kernel<" in n:
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
              "MfmaUtil_raw_gfx94x_formula_percent": round(100.0 * busy / (gui * 256 * 4), 3) if gui else None,
              "MfmaUtil_percent": round(100.0 * busy / (gui / 8 * 256 * 4), 2) if gui else None}
print(json.dumps(out, indent=1))
