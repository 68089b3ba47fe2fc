#!/usr/bin/env python3
"""Summarise a rocprofv3 results.db (rocpd sqlite): per-kernel launch count / avg / total duration, and
per-kernel PMC counter sums when the pass collected counters.  Usage: rocpd_summary.py <results.db> [top_n]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    c = sqlite3.connect(db)
    rows = c.execute("""
        select s.kernel_name, count(*), avg(d.end - d.start), sum(d.end - d.start), min(d.end-d.start), max(d.end-d.start),
               max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size)
        from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
        group by s.kernel_name order by 4 desc""").fetchall()
    tot = sum(r[3] for r in rows) or 1
    print(f"# {db}")
    print(f"{'kernel':<78} {'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'total_ms':>10} {'%':>6} vgpr agpr sgpr lds")
    for r in rows[:top]:
        name = r[0].split("(")[0][-76:]
        print(f"{name:<78} {r[1]:>6} {r[2]/1e3:>10.1f} {r[4]/1e3:>10.1f} {r[5]/1e3:>10.1f} {r[3]/1e6:>10.3f} {100*r[3]/tot:>6.2f} "
              f"{r[6]} {r[7]} {r[8]} {r[9]}")
    print(f"total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches, {len(rows)} distinct kernels")
    try:
        pm = c.execute("""
            select s.kernel_name, p.name, count(*), sum(e.value), avg(e.value)
            from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
            join rocpd_kernel_dispatch d on d.event_id = e.event_id
            join rocpd_info_kernel_symbol s on d.kernel_id = s.id
            group by s.kernel_name, p.name order by 4 desc""").fetchall()
        if pm:
            print("\n# PMC (per kernel: counter, dispatches, sum, avg per dispatch)")
            for r in pm[:top]:
                print(f"{r[0].split('(')[0][-70:]:<72} {r[1]:<14} {r[2]:>5} {r[3]:>18.1f} {r[4]:>16.1f}")
    except sqlite3.Error as e:
        print("no pmc:", e)


if __name__ == "__main__":
    main()
