#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table:
calls, total ms, avg us, % of GPU kernel time.

    rocpd_stats.py results.db [--top N] [--marker SUBSTR --last K]

With --marker, only the window spanning the last K intervals between consecutive dispatches of the marker kernel is
summarised (one marker dispatch per step -> K steady-state steps, excluding warm-up / MIOpen find), and per-step
figures are printed."""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--top", type=int, default=40)
ap.add_argument("--marker", default=None)
ap.add_argument("--last", type=int, default=3)
a = ap.parse_args()
db = sqlite3.connect(a.db)
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
nc = "name" if "name" in cols else "kernel_name"
where, nsteps = "", None
if a.marker:
    ts = [r[0] for r in cur.execute(f"select start from kernels where {nc} like ? order by start", (f"%{a.marker}%",))]
    assert len(ts) > a.last, f"only {len(ts)} marker dispatches"
    lo, hi = ts[-a.last - 1], ts[-1]
    where, nsteps = f"where start >= {lo} and start < {hi}", a.last
    print(f"window: last {a.last} steps delimited by '{a.marker}', wall {(hi - lo) / 1e6:.3f} ms = {(hi - lo) / 1e6 / a.last:.3f} ms/step")
rows = cur.execute(f"select {nc}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels {where} group by {nc} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
n = nsteps or 1
print(f"GPU kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches" + (f" = {tot / 1e6 / n:.3f} ms/step, {sum(r[1] for r in rows) // n} dispatches/step" if nsteps else ""))
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|")
for name, c, t, avg, mn, mx in rows[:a.top]:
    name = name if len(name) < 100 else name[:97] + "..."
    print(f"| {name} | {c} | {t / 1e6:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * t / tot:.1f} |")
