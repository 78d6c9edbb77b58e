"""Phase timing inside one k_wino workgroup (debug build of conv_wino.hip with -DCAGC_WINO_TRACE):
  cd content-aware-gan-compression_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCAGC_WINO_TRACE -c conv_wino.hip -o build/conv_wino_trace.o \
     && hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v conv_wino) build/conv_wino_trace.o -o ../cagc/libcagc_trace.so
  python scripts/trace_wino.py            (env B, C, H; CAGC_WINO_ORDER)"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc import _lib
_lib.LIB_PATH = os.path.join(ROOT, "content-aware-gan-compression_amd", "cagc", "libcagc_trace.so")
from cagc.op import modconv as mc
B, C, H = int(os.environ.get("B", 16)), int(os.environ.get("C", 512)), int(os.environ.get("H", 64))
x = torch.randn(B, C, H, H, device="cuda"); w = torch.randn(C, C, 3, 3, device="cuda")
up = mc.pack_wino(w, 0.01, False); out = torch.empty_like(x)
for _ in range(3):
    _lib.call("cagc_wino_conv3x3", _lib.ptr(out), _lib.ptr(x), _lib.ptr(up), None, B, C, C, H, H, 0, None, None, 0, None, None, 0.2, 1.0)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 64)()
assert lib.cagc_wino_trace_dump(buf) == 0
nch = (C + 7) // 8
names = ["commit+prefetch", "transform(before)", "multiply", "transform(after)", "barrier"]
print(f"B {B} C {C} H {H}: {nch} chunks; shader cycles per chunk, per wave (wave w and w+4 share a SIMD)")
for wv in range(8):
    row = [buf[wv * 8 + k] / nch for k in range(5)]
    print(f"wave {wv}: " + "  ".join(f"{n} {v:7.0f}" for n, v in zip(names, row)) + f"   total {sum(row):7.0f}"
          + f"   | whole kernel: prologue {buf[wv * 8 + 5]} main {sum(buf[wv * 8 + k] for k in range(5))} epilogue {buf[wv * 8 + 6]} cycles")
