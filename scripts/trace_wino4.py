"""Phase timing inside one k_wino4 workgroup (debug build of conv_wino4.hip with -DCAGC_W4_TRACE):
  scripts/build_wino4_variants.sh trace "-DCAGC_W4_TRACE=1"   ->  cagc/libcagc_hip_trace.so
  python scripts/trace_wino4.py            (env B, C, H; 128-channel workgroup shape forced)
Per wave (wave w and w + 4 share a SIMD and alternate as the chunk's transformer): shader cycles per chunk in each phase."""
import ctypes, os, sys, torch
os.environ["CAGC_WINO4_HV"] = "2"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import _lib
_lib.LIB_PATH = os.path.join(ROOT, "content-aware-gan-compression_amd", "cagc", os.environ.get("LIB", "libcagc_hip_trace.so"))
from cagc.op import modconv as mc
B, C, H = int(os.environ.get("B", 16)), int(os.environ.get("C", 512)), int(os.environ.get("H", 64))
x = torch.randn(B, C, H, H, device="cuda"); w = torch.randn(C, C, 3, 3, device="cuda")
up = mc.pack_wino(w, 0.01, False); out = torch.empty_like(x)
for _ in range(3):
    _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(x), _lib.ptr(up), None, B, C, C, H, H, 0, None, None, 0, None, None, 0.2, 1.0)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 96)()
assert lib.cagc_wino4_trace_dump(buf) == 0
nch = (C + 7) // 8
half = nch / 2.0     # every wave transforms in half of the chunks
print(f"B {B} C {C} H {H}: {nch} chunks of 8 channels, 18 MFMA groups (72 MFMAs per wave) per chunk; ideal with two waves per SIMD: 4608 cycles per chunk.")
print("| wave | g0-5 (6 groups) | g6 + commit | g7 | g8 when transforming | g8 when not | g9-17 (9 groups) | barrier after transforming | barrier after not | per chunk | prologue | epilogue |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
tot = [0.0] * 8
for wv in range(8):
    t = [buf[wv * 12 + k] for k in range(12)]
    row = [t[0] / nch, t[1] / nch, t[2] / nch, t[3] / half, t[8] / half, t[4] / nch, t[5] / half, t[9] / half]
    per = sum(t[k] for k in (0, 1, 2, 3, 8, 4, 5, 9)) / nch
    print(f"| {wv} | " + " | ".join(f"{v:.0f}" for v in row) + f" | {per:.0f} | {t[6]} | {t[7]} |")
    for i, v in enumerate(row): tot[i] += v / 8
print("| mean | " + " | ".join(f"{v:.0f}" for v in tot) + " | | | |")
print(f"per pure group (4 MFMAs of this wave, partner's 4 interleaved): g0-5 {tot[0] / 6:.0f}, g9-17 {tot[5] / 9:.0f} cycles (256 = the matrix pipe never idle)")
print(f"the transform adds {tot[3] - tot[2]:.0f} cycles to its group for the wave that runs it; its partner's same group takes {tot[4]:.0f} ({tot[4] - tot[2]:+.0f} vs a pure group)")
print(f"barrier wait: {tot[6]:.0f} cycles after transforming, {tot[7]:.0f} after not")
