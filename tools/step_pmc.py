#!/usr/bin/env python3
"""Whole-step matrix-pipe utilisation of the KD step from one rocprofv3 PMC pass.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv \
        -d gpurun_out/pmc_step -o s -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras
    python tools/step_pmc.py gpurun_out/pmc_step > profiles/r03_step_pmc.json

SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (cycles the matrix pipe executes: 64 per v_mfma_f32_32x32x2_f32),
GRBM_GUI_ACTIVE over the 8 XCDs.  Under --pmc rocprofv3 serialises the kernels (the teacher branch no longer overlaps the
student), so the denominator is the SUM of kernel-active cycles — utilisation while a kernel runs — and the per-kernel-family
table says where the idle matrix-pipe cycles are."""
import collections
import csv
import glob
import json
import re
import sys

SIMDS, XCDS = 1024, 8
per = collections.defaultdict(lambda: collections.Counter())
steps, seen = 0, set()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        raw = r["Kernel_Name"]
        m = re.search(r"(\w+)(<[^(]*>)?\(", raw.replace("(anonymous namespace)::", "").replace("void ", ""))
        name = m.group(1) if m else raw[:40]
        per[name][r["Counter_Name"]] += float(r["Counter_Value"])
        per[name]["n:" + r["Counter_Name"]] += 1
        if "token_kd_ce_kernel" in raw and r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            steps += 1
steps = max(steps, 1)
busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"] for c in per.values())
gui = sum(c["GRBM_GUI_ACTIVE"] for c in per.values())
rows = []
for k, c in sorted(per.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])[:14]:
    cyc = c["GRBM_GUI_ACTIVE"] / XCDS
    rows.append({"kernel": k, "launches_per_step": round(c["n:GRBM_GUI_ACTIVE"] / steps, 1),
                 "active_cycles_per_step": round(cyc / steps), "share_of_active_cycles": round(c["GRBM_GUI_ACTIVE"] / max(gui, 1), 4),
                 "mfma_busy": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMDS / max(cyc, 1), 4)})
print(json.dumps({
    "workload": "cfg3 KD step, B=64, fp32, hipGraph (bench.py --steps 4 --warmup 2 --no-extras), kernels serialised by the profiler",
    "method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace; whole-step figure = sum of "
              "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs divided by sum of GRBM_GUI_ACTIVE / 8 XCDs over every dispatch",
    "steps_seen": steps, "mfma_busy_cycles_per_simd_per_step": round(busy / SIMDS / steps),
    "kernel_active_cycles_per_step": round(gui / XCDS / steps),
    "whole_step_mfma_busy_fraction": round(busy / SIMDS / max(gui / XCDS, 1), 4), "by_kernel": rows}, indent=1))
