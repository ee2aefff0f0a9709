"""Reference point only (NOT used by the product): what the vendor BLAS behind torch.mm achieves on the skinny fp32 shapes,
measured the same way as tools/gemm_tune.py (rotating cold weights, back-to-back launches)."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
shapes = [(128, 5120, 1280), (128, 1280, 5120), (32, 5120, 1280), (32, 1280, 5120), (512, 2560, 640), (512, 640, 2560), (128, 3840, 1280),
          (128, 1280, 1280), (1024, 5120, 1280), (4096, 2560, 640), (2048, 8192, 256),
          (4096, 5120, 1280), (4096, 1280, 5120), (16384, 2560, 640), (32768, 5120, 1280), (32768, 1280, 5120), (131072, 2560, 640), (131072, 640, 2560)]
for M, N, K in shapes:
    ncopy = max(2, min(64, int(600e6 // (N * K * 4)) + 1)) if M < 4096 else 3
    A = torch.randn(M, K, device="cuda")
    Ws = [torch.randn(N, K, device="cuda") for _ in range(ncopy)]
    C = torch.empty(M, N, device="cuda")
    for W in Ws[:2]:
        torch.mm(A, W.t(), out=C)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for W in Ws:
            torch.mm(A, W.t(), out=C)
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / ncopy)
    us = sorted(ts)[2]
    print("%5d x %5d x %5d  vendor BLAS fp32: %7.1f us  %6.1f TF" % (M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
