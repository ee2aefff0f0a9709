"""Run one GEMM variant repeatedly (for rocprofv3 --pmc): python tools/gemm_one.py M N K cfg splitk [iters]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from paella_amd import _lib

M, N, K, cfg, sk = (int(v) for v in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
lib = _lib.load()
A = torch.randn(M, K, device="cuda")
W = torch.randn(N, K, device="cuda")
C = torch.empty(M, N, device="cuda")
ws = _lib.new_workspace(256 << 20, "cuda")
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(iters):
    flush.zero_()
    rc = lib.paella_op_gemm(A.data_ptr(), W.data_ptr(), None, None, C.data_ptr(), M, N, K, 0, cfg, sk, ws.data_ptr(), ws.numel(), st)
    assert rc == 0, lib.paella_last_error()
torch.cuda.synchronize()
