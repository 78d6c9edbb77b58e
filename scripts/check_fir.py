import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "content-aware-gan-compression_amd")]
from cagc.op.upfirdn2d import upfirdn2d
from oracle import ref_ops
k = torch.tensor([1., 3., 3., 1.]); k = (k[:, None] * k[None, :]); k = (k / k.sum()).cuda()
for (n, c, h, w, up, down, pad) in [(2, 3, 8, 8, 1, 2, (1, 1)), (2, 3, 8, 8, 2, 1, (2, 1)), (16, 128, 256, 256, 1, 2, (1, 1)),
                                    (16, 256, 128, 128, 2, 1, (2, 1)), (16, 3, 128, 128, 2, 1, (2, 1)), (16, 3, 256, 256, 1, 2, (1, 1)),
                                    (4, 5, 33, 47, 1, 2, (1, 1)), (4, 5, 33, 47, 2, 1, (2, 1)), (4, 5, 34, 46, 1, 2, (2, 1)), (1, 512, 8, 8, 1, 2, (1, 1))]:
    x = torch.randn(n, c, h, w, device="cuda")
    y = upfirdn2d(x, k, up=up, down=down, pad=pad)
    torch.cuda.synchronize()
    r = ref_ops.upfirdn2d_ref(x, k, up=up, down=down, pad=pad)
    print((n, c, h, w, up, down, pad), tuple(y.shape), "err %.2e" % ((y - r).abs().max() / r.abs().max()).item(), flush=True)
