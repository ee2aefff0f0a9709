import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from paella_amd import _lib
lib = _lib.load()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = _lib.new_workspace(256 << 20, "cuda")
for (M, N, K, sk) in [(128, 1280, 1280, -640), (128, 1280, 5120, -1280), (32, 1280, 1280, -160), (512, 640, 2560, -1280)]:
    ncopy = 24
    A = torch.randn(M, K, device="cuda")
    Ws = [torch.randn(N, K, device="cuda") for _ in range(ncopy)]
    Rs = [torch.randn(M, N, device="cuda") for _ in range(ncopy)]
    bias = torch.randn(N, device="cuda")
    C = torch.empty(M, N, device="cuda")
    def run(mode):
        ts = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for W, R in zip(Ws, Rs):
                lib.paella_op_gemm(A.data_ptr(), W.data_ptr(), bias.data_ptr() if mode else None, R.data_ptr() if mode == 2 else None, C.data_ptr(), M, N, K, 0, 30, sk, ws.data_ptr(), ws.numel(), st())
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / ncopy)
        ts.sort(); return ts[len(ts) // 2]
    print("%dx%dx%d ring30/%d: no epilogue operands %.2f us | + bias %.2f | + bias + cold residual %.2f" % (M, N, K, sk, run(0), run(1), run(2)))
