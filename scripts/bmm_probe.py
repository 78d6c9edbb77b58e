import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda"
for (C, K) in [(512, 16384), (512, 65536), (256, 65536), (128, 262144), (512, 4096)]:
    a = torch.randn(36, C, K, device=dev); b = torch.randn(36, C, K, device=dev)
    for _ in range(2): p = torch.bmm(a, b.transpose(1, 2))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): p = torch.bmm(a, b.transpose(1, 2))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    fl = 2.0 * 36 * C * C * K
    # exactness check vs float64 on a slice
    ref = (a[0, :8].double() @ b[0, :8].double().t())
    err = float((p[0, :8, :8].double() - ref).abs().max() / ref.abs().max())
    print(f"C {C} K {K}: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TF/s  rel err {err:.1e}", flush=True)
    del a, b, p
