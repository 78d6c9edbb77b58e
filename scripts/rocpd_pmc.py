#!/usr/bin/env python3
"""Per-kernel sums of one PMC counter from a rocprofv3 rocpd database (counters_collection view), restricted to the
last K marker-delimited steps like rocpd_stats.py.   rocpd_pmc.py results.db [--marker k_gan_kd_loss_tail --last 2] [--top 25]"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--marker", default=None)
ap.add_argument("--last", type=int, default=2)
ap.add_argument("--top", type=int, default=25)
ap.add_argument("--mfma", action="store_true", help="pivot the SQ MFMA / busy counters into one row per kernel with the busy fraction")
a = ap.parse_args()
cur = sqlite3.connect(a.db).cursor()
where = ""
if a.marker:
    ts = [r[0] for r in cur.execute("select start from counters_collection where kernel_name like ? group by dispatch_id order by start", (f"%{a.marker}%",))]
    assert len(ts) > a.last, f"only {len(ts)} marker dispatches"
    where = f"where start >= {ts[-a.last - 1]} and start < {ts[-1]}"
    print(f"window: last {a.last} steps delimited by '{a.marker}'")
rows = cur.execute(f"select kernel_name, counter_name, count(distinct dispatch_id), sum(value), sum(end-start) from counters_collection {where} group by kernel_name, counter_name order by 4 desc").fetchall()
print("| kernel | counter | dispatches | sum | per dispatch | total ms |")
print("|---|---|---|---|---|---|")
for n, c, k, v, t in rows[:a.top]:
    n = n if len(n) < 90 else n[:87] + "..."
    print(f"| {n} | {c} | {k} | {v:.4g} | {v / k:.4g} | {t / 1e6:.3f} |")

if a.mfma:
    # one row per kernel: MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the SQ counter
    # sums busy cycles over every SIMD of the chip (256 CUs x 4), GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles
    piv = {}
    for n, c, k, v, t in rows:
        piv.setdefault(n, {})[c] = (k, v, t)
    print()
    print("| kernel | dispatches | total ms | MFMA busy cycles / dispatch | GUI active cycles / dispatch | MFMA busy fraction | MFMA insts / dispatch |")
    print("|---|---|---|---|---|---|---|")
    order = sorted(piv.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", (0, 0, 0))[2])
    for n, d in order[:40]:
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "GRBM_GUI_ACTIVE" not in d:
            continue
        k, busy, t = d["SQ_VALU_MFMA_BUSY_CYCLES"]
        gui = d["GRBM_GUI_ACTIVE"][1]
        insts = d.get("SQ_INSTS_MFMA", (k, 0, 0))[1]
        if busy == 0:
            continue
        nm = n if len(n) < 90 else n[:87] + "..."
        print(f"| {nm} | {k} | {t / 1e6:.3f} | {busy / k:.4g} | {gui / k:.4g} | {busy / (gui * 128):.3f} | {insts / k:.4g} |")
