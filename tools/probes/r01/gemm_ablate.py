"""Ablation of the tiled GEMM main loop (guide: ablate before optimising): full vs no-MFMA vs no-global-load vs MFMA-only."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paella_amd import _lib
lib = _lib.load()
lib.paella_debug_set_spread(4)  # two-launch mode: the ablated kernels write slabs / outputs but no combine
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
def bench(M, N, K, cfg, sk, ncopy=24, reps=5):
    A = torch.randn(M, K, device="cuda"); C = torch.empty(M, N, device="cuda")
    Ws = [torch.randn(N, K, device="cuda") for _ in range(ncopy)]
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(ncopy):
            rc = lib.paella_op_gemm(A.data_ptr(), Ws[i].data_ptr(), None, None, C.data_ptr(), M, N, K, 0, cfg, sk, ws.data_ptr(), ws.numel(), st)
            assert rc == 0, lib.paella_last_error()
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / ncopy)
    return sorted(ts)[len(ts) // 2]
for (M, N, K, tile, sk) in [(128, 5120, 1280, 5, 1), (128, 5120, 1280, 5, 2), (128, 5120, 1280, 2, 4), (1024, 5120, 1280, 2, 1), (32, 5120, 1280, 5, 4)]:
    full = bench(M, N, K, tile, sk)
    r = [bench(M, N, K, 64 + 16 * a + tile, sk) for a in (1, 2, 3)]
    print("%5dx%5dx%5d tile %d S=%d: full %.1f us | no-MFMA %.1f | no-global-load/LDS-store %.1f | MFMA-only(no frag reads) %.1f" % (M, N, K, tile, sk, full, r[0], r[1], r[2]), flush=True)
