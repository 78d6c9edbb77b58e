"""Workgroup-level timeline of ONE register-direct conv launch (debug build -DCAGC_RD_TRACE: every workgroup records start / end in
100 MHz ticks, its item and its XCC).  Build (from content-aware-gan-compression_amd/csrc, after `make`):
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DCAGC_RD_TRACE=1 -c conv_rd.hip -o build_alt/conv_rd_trace.o
  hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v "build/conv_rd.o") build_alt/conv_rd_trace.o -o ../cagc/libcagc_hip_rdtrace.so
Run ON THE GPU BOX:  LIB=libcagc_hip_rdtrace.so SHAPE=cin,cout,H python scripts/trace_rd.py
Prints per item kind (taps) the count and duration, the number of busy workgroup slots over time, and what the tail costs."""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import _lib
_lib.LIB_PATH = os.path.join(ROOT, "content-aware-gan-compression_amd", "cagc", os.environ.get("LIB", "libcagc_hip_rdtrace.so"))
from cagc.op import modconv as mc
B = int(os.environ.get("BS", "16"))
cin, cout, H = (int(v) for v in os.environ.get("SHAPE", "256,512,128").split(","))
hb = H + 1; pitch = (hb + 3) // 4 * 4; ho = (hb - 3) // 2 + 1
w = torch.randn(cout, cin, 3, 3, device="cuda")
wp_fwd, wp_bwd = mc.pack_plain_weights(w, 0.01, True)
g = torch.randn(B, cout, ho, ho, device="cuda"); gtmp = torch.empty(B, cin, hb, pitch, device="cuda")
run = lambda: _lib.call("cagc_conv3x3s2_dgrad", _lib.ptr(gtmp), _lib.ptr(g), _lib.ptr(wp_bwd), B, cin, cout, hb, hb, pitch)
for _ in range(3): run()
NMAX = 1 << 16
tr = torch.zeros(NMAX * 4, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
_lib.load().cagc_set_clock_probe(ctypes.c_void_p(tr.data_ptr()))
run(); torch.cuda.synchronize()
_lib.load().cagc_set_clock_probe(None)
t = tr.view(NMAX, 4).cpu()
t = t[t[:, 1] > 0]
n = t.shape[0]
t0 = int(t[:, 0].min()); start = (t[:, 0] - t0).double() / 100.0; end = (t[:, 1] - t0).double() / 100.0     # microseconds
dur = end - start; taps = t[:, 2] % 16; item = t[:, 2] // 16
total = float(end.max())
print(f"stride-2 data gradient {cout}->{cin} @{H}^2 batch {B}: {n} workgroups, launch {total:.1f} us, sum of workgroup time {float(dur.sum())/1e3:.2f} ms"
      f" = {float(dur.sum())/total:.1f} slots busy on average (of 512)")
print("| item | taps | workgroups | mean us | min | max | us per tap | first start | last end |")
print("|---|---|---|---|---|---|---|---|---|")
for it in sorted(set(item.tolist())):
    m = item == it; k = int(taps[m][0])
    print(f"| {it} | {k} | {int(m.sum())} | {float(dur[m].mean()):.1f} | {float(dur[m].min()):.1f} | {float(dur[m].max()):.1f} | {float(dur[m].mean())/k:.1f} |"
          f" {float(start[m].min()):.1f} | {float(end[m].max()):.1f} |")
# busy slots over time, 20 bins
import math
nb = 20
print("busy workgroup slots by time bin (%d bins of %.1f us):" % (nb, total / nb))
row = []
for i in range(nb):
    a, b = total * i / nb, total * (i + 1) / nb
    ov = (torch.minimum(end, torch.tensor(b)) - torch.maximum(start, torch.tensor(a))).clamp(min=0).sum() / (b - a)
    row.append(f"{float(ov):.0f}")
print(" ".join(row))
# what a perfectly packed launch of the same workgroups would take
ideal = float(dur.sum()) / 512.0
print(f"same workgroup durations packed perfectly on 512 slots: {ideal:.1f} us -> the schedule costs {100 * (total / ideal - 1):.1f} % of the launch")
x = t[:, 3]
print("workgroups per XCC:", {int(k): int((x == k).sum()) for k in sorted(set(x.tolist()))})
