import os, sys, time, torch
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.join(ROOT, "content-aware-gan-compression_amd", "cagc", sys.argv[1])
from cagc.op import modconv as mc
for (B, C, H) in [(16, 512, 64), (16, 512, 64), (16, 128, 256)]:
    x = torch.randn(B, C, H, H, device="cuda"); w = torch.randn(C, C, 3, 3, device="cuda")
    up = mc.pack_wino(w, 0.01, False); out = torch.empty_like(x)
    def run(): _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(x), _lib.ptr(up), None, B, C, C, H, H, 0, None, None, 0, None, None, 0.2, 1.0)
    for _ in range(3): run()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print(sys.argv[1:] or "default", (B, C, H), f"{dt*1e3:.3f} ms", flush=True)
