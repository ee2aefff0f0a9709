"""Fixed cost vs per-k-tile slope of the fp32 GEMM: time M x N x K for growing K at fixed tile config / split-K.
Usage (GPU box): python tools/gemm_kscale.py [M N [cfg/splitk,cfg/splitk,...]]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from paella_amd import _lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5120
lib = _lib.load()
ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
print("M=%d N=%d; us per launch (back-to-back, rotating cold weights)" % (M, N))
VARIANTS = [tuple(int(x) for x in v.split("/")) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else ((5, 1), (5, 2), (5, 4), (2, 1), (2, 4), (3, 2), (4, 2))
for cfg, sk in VARIANTS:
    out = []
    for K in (32, 64, 128, 256, 512, 1280, 2560, 5120):
        if K // sk < 32:
            out.append("   -  ")
            continue
        ncopy = max(2, min(64, int(600e6 // (N * K * 4)) + 1))
        A = torch.randn(M, K, device="cuda")
        Ws = [torch.randn(N, K, device="cuda") for _ in range(ncopy)]
        C = torch.empty(M, N, device="cuda")
        run = lambda W: lib.paella_op_gemm(A.data_ptr(), W.data_ptr(), None, None, C.data_ptr(), M, N, K, 0, cfg, sk, ws.data_ptr(), ws.numel(), st())
        assert run(Ws[0]) == 0
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for W in Ws:
                run(W)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / ncopy)
        ts.sort()
        out.append("%6.1f" % ts[2])
        del Ws
    print("cfg %d S=%d  K=32,64,128,256,512,1280,2560,5120: %s" % (cfg, sk, " ".join(out)), flush=True)
