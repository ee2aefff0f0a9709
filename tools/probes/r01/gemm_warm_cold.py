"""A/B: same GEMM with warm weights (one W, MALL/TLB resident) vs cold weights (rotation > MALL)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paella_amd import _lib
lib = _lib.load()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
def bench(M, N, K, cfg, sk, ncopy, reps=5):
    A = torch.randn(M, K, device="cuda"); C = torch.empty(M, N, device="cuda")
    Ws = [torch.randn(N, K, device="cuda") for _ in range(ncopy)]
    n = max(ncopy, 24)
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            lib.paella_op_gemm(A.data_ptr(), Ws[i % ncopy].data_ptr(), None, None, C.data_ptr(), M, N, K, 0, cfg, sk, ws.data_ptr(), ws.numel(), st)
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return sorted(ts)[len(ts) // 2]
for (M, N, K, cfg, sk) in [(128, 5120, 1280, 3, 2), (128, 5120, 1280, 4, 4), (32, 1280, 5120, 5, 16), (128, 1280, 1280, 5, 4), (512, 640, 2560, 5, 4), (128, 5120, 1280, 5, 1), (128, 5120, 1280, 2, 4), (32, 5120, 1280, 5, 4), (128, 1280, 5120, 5, 8), (512, 2560, 640, 5, 1), (1024, 5120, 1280, 2, 1)]:
    lib.paella_debug_set_spread(0)
    off = bench(M, N, K, cfg, sk, 28)
    lib.paella_debug_set_spread(2)
    on = bench(M, N, K, cfg, sk, 28)
    print("%5d x %5d x %5d cfg %d/%d: register ring %.1f us | glds LDS ring %.1f us" % (M, N, K, cfg, sk, off, on), flush=True)
