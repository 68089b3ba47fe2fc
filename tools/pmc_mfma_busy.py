#!/usr/bin/env python3
"""Per-kernel matrix-core occupancy from rocprofv3 PMC passes (rocpd sqlite): for every kernel of every database given,
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip, per dispatch) / (1024 SIMDs x GRBM_GUI_ACTIVE cycles per dispatch)
(SQ_VALU_MFMA_BUSY_CYCLES counts cycles: 32 per v_mfma_f32_32x32x16_{f16,bf16} -- MI355X_MICROARCH.md; GRBM_GUI_ACTIVE = shader-clock
cycles the dispatch was resident), plus every other counter as its per-dispatch chip-wide sum and, for the SQ quad-cycle counters, as a
fraction of SQ_WAVE_CYCLES.  Usage: pmc_mfma_busy.py out.json <results.db> [<results.db> ...]   (one db per --pmc pass)"""
import json
import sqlite3
import sys


def load(db, acc):
    c = sqlite3.connect(db)
    disp = c.execute("""select s.kernel_name, count(*), avg(d.end - d.start) from rocpd_kernel_dispatch d
                        join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name""").fetchall()
    for name, n, avg_ns in disp:
        k = acc.setdefault(name.split("(")[0], {"counters": {}})
        k["dispatches"], k["avg_us_under_pmc"] = n, round(avg_ns / 1e3, 1)
    rows = c.execute("""select s.kernel_name, p.name, count(*), sum(e.value), count(distinct d.id)
                        from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
                        join rocpd_kernel_dispatch d on d.event_id = e.event_id
                        join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name, p.name""").fetchall()
    for name, ctr, n_ev, total, n_disp in rows:
        k = acc.setdefault(name.split("(")[0], {"counters": {}})
        k["counters"][ctr] = {"sum_per_dispatch": total / max(1, n_disp), "instances_per_dispatch": n_ev / max(1, n_disp)}


def main():
    out, dbs = sys.argv[1], sys.argv[2:]
    acc = {}
    for db in dbs:
        load(db, acc)
    res = {}
    for name, k in acc.items():
        ctr = k["counters"]
        if not ctr:
            continue
        r = {"dispatches": k.get("dispatches"), "avg_us_under_pmc": k.get("avg_us_under_pmc")}
        gui = ctr.get("GRBM_GUI_ACTIVE")
        cycles = gui["sum_per_dispatch"] / gui["instances_per_dispatch"] if gui else None     # every instance reports the same interval
        if cycles:
            r["gpu_cycles_per_dispatch"] = round(cycles)
            r["effective_clock_GHz"] = round(cycles / (k["avg_us_under_pmc"] * 1e3), 3) if k.get("avg_us_under_pmc") else None
        mf = ctr.get("SQ_VALU_MFMA_BUSY_CYCLES")
        if mf and cycles:
            r["mfma_busy"] = round(mf["sum_per_dispatch"] / (1024.0 * cycles), 4)
        wave = ctr.get("SQ_WAVE_CYCLES")
        for c, v in sorted(ctr.items()):
            r[c] = round(v["sum_per_dispatch"], 1)
            if wave and c.startswith("SQ_") and c not in ("SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES") and not c.startswith("SQ_INSTS"):
                r[c + "_over_wave_cycles"] = round(v["sum_per_dispatch"] / wave["sum_per_dispatch"], 4)
        hit, miss = ctr.get("TCC_HIT_sum"), ctr.get("TCC_MISS_sum")
        if hit and miss and hit["sum_per_dispatch"] + miss["sum_per_dispatch"] > 0:
            r["l2_hit_rate"] = round(hit["sum_per_dispatch"] / (hit["sum_per_dispatch"] + miss["sum_per_dispatch"]), 4)
        res[name] = r
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    for name, r in sorted(res.items(), key=lambda kv: -(kv[1].get("avg_us_under_pmc") or 0) * (kv[1].get("dispatches") or 0))[:16]:
        print("%-72s %8.1f us  mfma_busy %s  lds_wait %s  l2_hit %s" % (name[-72:], r.get("avg_us_under_pmc") or 0, r.get("mfma_busy"),
                                                                     r.get("SQ_WAIT_INST_LDS_over_wave_cycles"), r.get("l2_hit_rate")))


if __name__ == "__main__":
    main()
