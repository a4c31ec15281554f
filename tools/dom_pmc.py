#!/usr/bin/env python3
"""Dominant-kernel PMC summary from `rocprofv3 --pmc ... -- python3 bench.py --dominant-kernel-only` (tools/prof_r03.sh):
   python tools/dom_pmc.py gpurun_out/pmc_dom gpurun_out/prof_dom.json > profiles/r03_dominant_kernel_pmc.json
Averages per launch over the launches of the kernel with the largest total GRBM_GUI_ACTIVE (the three candidates of
bench.dominant_kernel are all in the trace)."""
import collections
import csv
import glob
import json
import sys

per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "igemm" not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"], r.get("Grid_Size", ""))
        per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
best = max(per, key=lambda k: sum(per[k]["GRBM_GUI_ACTIVE"]))
c = per[best]
avg = {k: sum(v) / len(v) for k, v in c.items()}
cyc = avg["GRBM_GUI_ACTIVE"] / 8
busy = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024
bench = {}
try:
    bench = json.load(open(sys.argv[2]))["dominant_kernel"]
except Exception:
    pass
print(json.dumps({
    "command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS "
               "SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -- python3 bench.py --dominant-kernel-only",
    "kernel": best[0], "grid": best[1], "launches": len(c["GRBM_GUI_ACTIVE"]), "bench_line": bench,
    "averages_per_launch": {k: round(v, 1) for k, v in avg.items()},
    "derived": {"kernel_cycles_per_xcd": round(cyc), "mfma_busy_cycles_per_simd": round(busy),
                "mfma_pipe_utilisation": round(busy / cyc, 4),
                "lds_bank_conflict_fraction": round(avg.get("SQ_LDS_BANK_CONFLICT", 0) / max(avg.get("SQ_LDS_IDX_ACTIVE", 1), 1), 4)},
    "notes": "GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (64 cycles per v_mfma_f32_32x32x2_f32)"},
    indent=1))
