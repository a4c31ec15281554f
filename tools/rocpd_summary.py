#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 rocpd database (kernel-trace): usage rocpd_summary.py results.db [launches_per_name_divisor]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
N = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = db.execute("select name, duration from kernels").fetchall()


def short(n):
    m = re.search(r'igemm_glds_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>', n)
    if m:
        return 'glds op%s %sx%s nb%s T%s w%s' % m.groups()
    m = re.search(r'(\w+)<([^(]*)>\(', n)
    if m:
        return m.group(1) + '<' + m.group(2)[:44] + '>'
    m = re.search(r'(\w+)\(', n)
    return m.group(1) if m else n[:60]


agg = collections.defaultdict(lambda: [0, 0])
for n, d in rows:
    agg[short(n)][0] += d
    agg[short(n)][1] += 1
tot = sum(v[0] for v in agg.values())
print(f"total kernel time {tot / N / 1e6:.3f} ms per step (divisor {N})")
for n, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
    print(f"{t / N / 1e3:9.1f} us/step {c / N:6.1f}/step {t / c / 1e3:8.1f} us  {n}")
