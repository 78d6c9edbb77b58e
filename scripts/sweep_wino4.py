import os, sys, time, torch
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import _lib
from cagc.op import modconv as mc
dev = "cuda"
SHAPES = [(16, 512, 64), (16, 512, 64), (16, 512, 32), (16, 256, 128), (16, 128, 256), (8, 512, 32), (4, 512, 32), (2, 512, 32), (4, 512, 64), (2, 512, 64), (2, 256, 128), (2, 128, 256), (16, 154, 64), (16, 154, 32), (2, 154, 64)]
for (B, C, H) in SHAPES:
    x = torch.randn(B, C, H, H, device=dev); w = torch.randn(C, C, 3, 3, device=dev)
    up = mc.pack_wino(w, 0.01, False); out = torch.empty_like(x)
    def run():
        _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(x), _lib.ptr(up), None, B, C, C, H, H, 0, None, None, 0, None, None, 0.2, 1.0)
    for _ in range(3): run()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    fl = 2.0 * B * C * C * 9 * H * H
    print(f"F4={os.environ.get('CAGC_WINO_F4', '1')} hv={os.environ.get('CAGC_WINO4_HV', 'auto')} B{B} C{C} H{H}: {dt*1e3:.3f} ms  direct-equiv {fl/dt/1e12:.1f} TF", flush=True)
