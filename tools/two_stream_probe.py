"""Probe: does running the cond / uncond halves as two concurrent streams beat one 2x-row stream for skinny GEMM chains?"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paella_amd import _lib
lib = _lib.load()
ws1 = _lib.new_workspace(128 << 20, "cuda")
ws2 = _lib.new_workspace(128 << 20, "cuda")
def chain(M, N, K, Ws, ws, stream):
    A = chain.A[(M, K)]; C = chain.C[(M, N)]
    sp = ctypes.c_void_p(stream.cuda_stream)
    for W in Ws:
        lib.paella_op_gemm(A.data_ptr(), W.data_ptr(), None, None, C.data_ptr(), M, N, K, 0, -1, 1, ws.data_ptr(), ws.numel(), sp)
chain.A, chain.C = {}, {}
for (M, N, K) in [(128, 5120, 1280), (128, 1280, 5120), (32, 5120, 1280), (32, 1280, 5120), (512, 2560, 640)]:
    n = 24
    Ws = [torch.randn(N, K, device="cuda") for _ in range(n)]
    for m in (M, M // 2):
        chain.A[(m, K)] = torch.randn(m, K, device="cuda"); chain.C[(m, N)] = torch.empty(m, N, device="cuda")
        chain.A[(m, K, 2)] = torch.randn(m, K, device="cuda")
    s0, s1, s2 = torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream()
    def one():
        chain(M, N, K, Ws, ws1, s0)
    def two():
        s1.wait_stream(s0); s2.wait_stream(s0)
        chain(M // 2, N, K, Ws, ws1, s1)
        chain(M // 2, N, K, Ws, ws2, s2)
        s0.wait_stream(s1); s0.wait_stream(s2)
    res = []
    for fn in (one, two):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / n)
        res.append(sorted(ts)[2])
    print("%4d x %5d x %5d: one stream (M rows) %.1f us/GEMM | two streams (M/2 rows each, same C buffer per stream) %.1f us per pair" % (M, N, K, res[0], res[1]), flush=True)
