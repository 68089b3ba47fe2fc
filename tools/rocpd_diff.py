#!/usr/bin/env python3
"""Per-kernel total-time difference between two rocprofv3 results.db files.  Usage: rocpd_diff.py a.db b.db [top]"""
import sqlite3
import sys


def load(db):
    c = sqlite3.connect(db)
    return {r[0]: (r[1], r[2]) for r in c.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
        "on d.kernel_id = s.id group by s.kernel_name")}


a, b = load(sys.argv[1]), load(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 15
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0))
    cb, tb = b.get(k, (0, 0))
    rows.append(((ta - tb) / 1e6, ca, cb, ta / 1e6, tb / 1e6, k.split("(")[0][-70:]))
rows.sort(key=lambda r: -abs(r[0]))
print(f"{'delta_ms':>9} {'calls_a':>7} {'calls_b':>7} {'a_ms':>9} {'b_ms':>9}  kernel")
for r in rows[:top]:
    print(f"{r[0]:>9.3f} {r[1]:>7} {r[2]:>7} {r[3]:>9.3f} {r[4]:>9.3f}  {r[5]}")
