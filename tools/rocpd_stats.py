#!/usr/bin/env python3
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as CSV.
    python tools/rocpd_stats.py gpurun_out/prof_x/x_results.db > profiles/r01x_kernel_stats.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print('"Name","Calls","TotalDurationUs","AverageUs","Percentage"')
for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    print(f'"{name}",{calls},{total:.2f},{avg:.3f},{pct:.3f}')
