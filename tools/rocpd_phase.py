#!/usr/bin/env python3
"""Concurrency of one family of kernels inside a step, from a rocprofv3 results.db: for the dispatches whose name matches <pattern>
between two occurrences of <marker> -- wall span (first start .. last end), union busy time, per-stream busy time and dispatch count,
the average number of matching kernels in flight, and what else ran inside the span.
Usage: rocpd_phase.py <results.db> <pattern> <marker> <occ_from> <occ_to>"""
import collections
import sqlite3
import sys


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main():
    db, pat, marker, o0, o1 = sys.argv[1], sys.argv[2].lower(), sys.argv[3].lower(), int(sys.argv[4]), int(sys.argv[5])
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)").fetchall()]
    scol = "d.stream_id" if "stream_id" in cols else "0"
    qcol = "d.queue_id" if "queue_id" in cols else "0"
    rows = c.execute("select d.start, d.end, s.kernel_name, %s, %s from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                     "on d.kernel_id = s.id order by d.start" % (scol, qcol)).fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[2].lower()]
    lo, hi = marks[o0], marks[o1]
    win = rows[lo + 1:hi]
    step_span = (rows[hi][0] - rows[lo][1]) / 1e3
    hit = [r for r in win if pat in r[2].lower()]
    if not hit:
        print("no match for", pat)
        return
    s0, e1 = min(r[0] for r in hit), max(r[1] for r in hit)
    print("# step window %.1f us, %d dispatches; '%s': %d dispatches" % (step_span, len(win), pat, len(hit)))
    print("span %.1f us   union busy %.1f us   sum of durations %.1f us   mean in flight %.2f" %
          ((e1 - s0) / 1e3, union([(r[0], r[1]) for r in hit]) / 1e3, sum(r[1] - r[0] for r in hit) / 1e3,
           sum(r[1] - r[0] for r in hit) / max(1, e1 - s0)))
    by = collections.defaultdict(list)
    for r in hit:
        by[(r[3], r[4])].append(r)
    for k, v in sorted(by.items()):
        print("  stream %s queue %s: %4d dispatches, first start +%.1f us, last end +%.1f us, busy %.1f us" %
              (k[0], k[1], len(v), (v[0][0] - s0) / 1e3, (max(r[1] for r in v) - s0) / 1e3, sum(r[1] - r[0] for r in v) / 1e3))
    other = [r for r in win if pat not in r[2].lower() and r[1] > s0 and r[0] < e1]
    print("other kernels inside the span: %d dispatches, %.1f us" % (len(other), sum(min(r[1], e1) - max(r[0], s0) for r in other) / 1e3))
    agg = collections.Counter()
    for r in other:
        agg[r[2].split("(")[0][-70:]] += (min(r[1], e1) - max(r[0], s0)) / 1e3
    for n, t in agg.most_common(8):
        print("    %9.1f us  %s" % (t, n))
    # the matching kernels by name
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in hit:
        a = agg[r[2].split("(")[0][-90:]]
        a[0] += 1
        a[1] += (r[1] - r[0]) / 1e3
    for n, (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print("  %5d x  %9.1f us  %s" % (cnt, t, n))


if __name__ == "__main__":
    main()
