#!/usr/bin/env python3
"""Kernel timeline from a rocprofv3 results.db (rocpd sqlite): dispatches in start order with the idle gap before each, per queue /
stream, restricted to a window around the dispatches whose name matches a pattern.
Usage: rocpd_timeline.py <results.db> <name-pattern> [before] [after] [occurrence (-1 = last)]"""
import sqlite3
import sys


def main():
    db, pat = sys.argv[1], sys.argv[2]
    before = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    after = int(sys.argv[4]) if len(sys.argv) > 4 else 12
    occ = int(sys.argv[5]) if len(sys.argv) > 5 else -1
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)").fetchall()]
    qcol = "queue_id" if "queue_id" in cols else None
    scol = "stream_id" if "stream_id" in cols else None
    sel = "d.start, d.end, s.kernel_name" + (", d.%s" % qcol if qcol else ", 0") + (", d.%s" % scol if scol else ", 0")
    rows = c.execute("select %s from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start" % sel).fetchall()
    hits = [i for i, r in enumerate(rows) if pat.lower() in r[2].lower()]
    print("# %d dispatches, %d match '%s'" % (len(rows), len(hits), pat))
    if not hits:
        names = sorted({r[2].split("(")[0][:90] for r in rows})
        print("\n".join(names))
        return
    i0 = hits[occ]
    lo, hi = max(0, i0 - before), min(len(rows), i0 + after + 1)
    t0 = rows[lo][0]
    last_end = rows[lo - 1][1] if lo > 0 else rows[lo][0]
    print(f"{'start_us':>10} {'dur_us':>9} {'gap_us':>9} {'q':>4} {'s':>4}  kernel")
    for i in range(lo, hi):
        st, en, name, q, s = rows[i]
        gap = (st - last_end) / 1e3
        print(f"{(st - t0) / 1e3:>10.1f} {(en - st) / 1e3:>9.1f} {gap:>9.1f} {q:>4} {s:>4}  {'>>' if i == i0 else '  '}{name.split('(')[0][-80:]}")
        last_end = max(last_end, en)


if __name__ == "__main__":
    main()
