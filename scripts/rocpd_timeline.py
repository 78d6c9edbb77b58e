#!/usr/bin/env python3
"""Timeline view of a rocprofv3 kernel-trace database (rocpd sqlite): for the last K steps (delimited by a marker kernel)
the wall time per step, the time at least one kernel is running (union of the dispatch intervals), the idle time between
kernels, the average number of kernels in flight, and where the idle time sits (largest gaps with the kernels around them) —
the view the small-per-GPU-batch work needs: a HIP-graph replay is bound by the dependency chain's gaps, not by kernel sums.

    rocpd_timeline.py results.db --marker k_gan_kd_loss_tail --last 4 [--gaps 25] [--chain]"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--marker", default="k_gan_kd_loss_tail")
ap.add_argument("--last", type=int, default=4)
ap.add_argument("--gaps", type=int, default=25)
ap.add_argument("--dump", action="store_true", help="print every dispatch of the LAST step: start offset, duration, name")
a = ap.parse_args()
db = sqlite3.connect(a.db)
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
nc = "name" if "name" in cols else "kernel_name"
ts = [r[0] for r in cur.execute(f"select start from kernels where {nc} like ? order by start", (f"%{a.marker}%",))]
assert len(ts) > a.last, f"only {len(ts)} marker dispatches"
lo, hi = ts[-a.last - 1], ts[-1]
rows = cur.execute(f"select start, end, {nc} from kernels where start >= {lo} and start < {hi} order by start").fetchall()
wall = hi - lo
busy, cur_end, ksum = 0, lo, 0
gaps = []
prev = None
for s, e, n in rows:
    ksum += e - s
    if s > cur_end:
        gaps.append((s - cur_end, prev, n, cur_end - lo))
        busy += e - s
        cur_end = e
    else:
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
    prev = n
short = lambda n: (n.split("(")[0])[:70]
print(f"window: last {a.last} steps delimited by '{a.marker}': wall {wall / 1e6 / a.last:.3f} ms/step, {len(rows) // a.last} dispatches/step")
print(f"kernel-time sum {ksum / 1e6 / a.last:.3f} ms/step; >= 1 kernel running {busy / 1e6 / a.last:.3f} ms/step; idle {(wall - busy) / 1e6 / a.last:.3f} ms/step "
      f"in {len(gaps) // a.last} gaps/step (mean {(wall - busy) / max(1, len(gaps)) / 1e3:.2f} us); kernels in flight while busy {ksum / max(1, busy):.2f}")
hist = {}
for g, _, _, _ in gaps:
    b = 1 if g < 1000 else 2 if g < 2000 else 4 if g < 4000 else 8 if g < 8000 else 16 if g < 16000 else 99
    hist[b] = hist.get(b, [0, 0])
    hist[b][0] += 1
    hist[b][1] += g
print("gap histogram (per step): " + ", ".join(f"<{b} us: {c / a.last:.0f} gaps / {t / 1e6 / a.last:.3f} ms" if b < 99 else f">=16 us: {c / a.last:.0f} / {t / 1e6 / a.last:.3f} ms"
                                                 for b, (c, t) in sorted(hist.items())))
print(f"\n| gap us | after kernel | before kernel |")
print("|---|---|---|")
for g, p, n, _ in sorted(gaps, key=lambda x: -x[0])[:a.gaps]:
    print(f"| {g / 1e3:.1f} | {short(p) if p else '-'} | {short(n)} |")
if a.dump:
    lo2 = ts[-2]
    print("\n| t us | dur us | kernel |")
    print("|---|---|---|")
    for s, e, n in rows:
        if s >= lo2:
            print(f"| {(s - lo2) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {short(n)} |")
