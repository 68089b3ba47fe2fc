#!/usr/bin/env python3
"""Idle gaps of the device between two occurrences of a marker kernel in a rocprofv3 results.db: total span, busy time, and every gap
longer than `min_gap_us` with the kernels on either side.  Usage: rocpd_gaps.py <results.db> <marker> <occ_from> <occ_to> [min_gap_us]"""
import sqlite3
import sys


def main():
    db, pat, o0, o1 = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    min_gap = float(sys.argv[5]) if len(sys.argv) > 5 else 50.0
    c = sqlite3.connect(db)
    rows = c.execute("select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                     "on d.kernel_id = s.id order by d.start").fetchall()
    hits = [i for i, r in enumerate(rows) if pat.lower() in r[2].lower()]
    a, b = hits[o0], hits[o1]
    span = (rows[b][0] - rows[a][0]) / 1e3
    busy, last_end, gaps = 0.0, rows[a][1], []
    for i in range(a + 1, b + 1):
        st, en, name = rows[i]
        if st > last_end:
            g = (st - last_end) / 1e3
            if g >= min_gap:
                gaps.append((g, (st - rows[a][0]) / 1e3, rows[i - 1][2].split("(")[0][-50:], name.split("(")[0][-50:]))
        busy += max(0, en - max(st, last_end)) / 1e3
        last_end = max(last_end, en)
    print(f"# {pat} occurrence {o0} -> {o1}: span {span:.1f} us, busy {busy:.1f} us, idle {span - busy:.1f} us, {b - a} dispatches")
    for g, at, p, n in gaps:
        print(f"gap {g:9.1f} us at +{at:10.1f} us   after {p}   before {n}")


if __name__ == "__main__":
    main()
