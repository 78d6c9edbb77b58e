import os, sys, ctypes, torch, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.join(ROOT, "content-aware-gan-compression_amd", "cagc", sys.argv[1])
from cagc.op import modconv as mc
B, C, H = int(os.environ.get("B", 16)), int(os.environ.get("C", 512)), int(os.environ.get("H", 64))
x = torch.randn(B, C, H, H, device="cuda"); w = torch.randn(C, C, 3, 3, device="cuda")
if os.environ.get("ZERO") == "1": x.zero_(); w.zero_()     # same instruction stream on zero operands: what the matrix pipe's power limit costs
up = mc.pack_wino(w, 0.01, False); out = torch.empty_like(x)
def run():
    _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(x), _lib.ptr(up), None, B, C, C, H, H, 0, None, None, 0, None, None, 0.2, 1.0)
for _ in range(3): run()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
fl = 2.0 * B * C * C * 9 * H * H
clk = torch.zeros(2, device="cuda")
_lib.load().cagc_set_clock_probe(ctypes.c_void_p(clk.data_ptr()))
for _ in range(5): run()
torch.cuda.synchronize(); _lib.load().cagc_set_clock_probe(None)
plan = _lib.query("cagc_wino_plan", B, C, C, H, H)
print(f"shader clock under this kernel: {float(clk[0] / clk[1].clamp(min=1)):.0f} MHz")
print(sys.argv[1:] or "default", (B, C, H), f"F({plan}x{plan}) ks_launches {_lib.get_tuning('wino4_ks_launches')}", f"{dt*1e3:.3f} ms  direct-equiv {fl/dt/1e12:.1f} TF  mfma {fl*4/9/dt/1e12:.1f} TF")
