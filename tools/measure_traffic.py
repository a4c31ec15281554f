#!/usr/bin/env python3
"""HBM bytes per KD step from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC slots).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras
    python tools/measure_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r01_traffic.json

Counter values are KB; FETCH_SIZE is doubled (MI355X_MICROARCH.md §HBM: on gfx950 it reports half the bytes of wide
coalesced reads, and every streaming read of this step is 16 B/lane, LDS-DMA included); steps executed are counted
from a once-per-step kernel (token_kd_ce_kernel)."""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    per, steps = collections.Counter(), 0
    seen = set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            raw = r["Kernel_Name"]
            name = raw.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            per[name] += float(r["Counter_Value"])
            if "token_kd_ce_kernel" in raw and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                steps += 1
    return per, steps


fetch, sf = load(sys.argv[1], "FETCH_SIZE")
write, sw = load(sys.argv[2], "WRITE_SIZE")
fb = sum(fetch.values()) * 1024 / max(sf, 1)
wb = sum(write.values()) * 1024 / max(sw, 1)
top = lambda c, s: {k: round(v / max(s, 1), 1) for k, v in c.most_common(12)}
print(json.dumps({
    "workload": "cfg3 KD step, B=64, fp32, hipGraph (bench.py --steps 6 --warmup 2)",
    "method": "rocprofv3 --pmc FETCH_SIZE --kernel-trace and, in a separate pass, --pmc WRITE_SIZE --kernel-trace; counter "
              "values are KB (x1024); summed over every kernel, divided by the steps executed; gfx950 correction: FETCH_SIZE x2",
    "steps_seen": [sf, sw], "fetch_bytes_raw_per_step": fb, "write_bytes_per_step": wb,
    "hbm_bytes_per_step_corrected": 2 * fb + wb,
    "per_kernel_fetch_kb_raw_per_step": top(fetch, sf), "per_kernel_write_kb_per_step": top(write, sw)}, indent=1))
