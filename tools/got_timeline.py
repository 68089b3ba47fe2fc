"""Kernel timeline of a rocprofv3 --kernel-trace database: start offset, duration, queue / stream and grid of every dispatch in a window
(used to see how the per-stain GOT chains of tools/exp_got_overlap.py overlap).  usage: got_timeline.py results.db [first_ms last_ms]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "queue_id" if "queue_id" in cols else None
scol = "stream_id" if "stream_id" in cols else None
sel = "d.start, d.end, s.kernel_name, d.grid_size_x" + (", d.%s" % qcol if qcol else ", 0") + (", d.%s" % scol if scol else ", 0")
rows = c.execute("select %s from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start" % sel).fetchall()
t0 = rows[0][0]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e18
for st, en, name, gx, q, sid in rows:
    ms = (st - t0) / 1e6
    if lo <= ms <= hi:
        short = name.split("(")[0][-46:]
        print("%10.3f ms  +%8.1f us  q%-3s s%-3s grid %-7d %s" % (ms, (en - st) / 1e3, q, sid, gx, short))
print("columns:", cols)
