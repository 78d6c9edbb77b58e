#!/usr/bin/env python3
"""Per-kernel sums of one PMC counter from a rocprofv3 rocpd database (counters_collection view), restricted to the
last K marker-delimited steps like rocpd_stats.py.   rocpd_pmc.py results.db [--marker k_masked_l1 --last 2] [--top 25]"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--marker", default=None)
ap.add_argument("--last", type=int, default=2)
ap.add_argument("--top", type=int, default=25)
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
