"""Why do the large GEMMs run ~5-10 % slower inside the model than in the isolated sweeps?  Times each shape (a) back to back with warm operands as
tools/gemm_tune.py does and (b) with a 1 GiB memset between launches (every operand cold in L2 and the Infinity Cache, as after the model's previous
kernels), event-timed around the GEMM alone.  Usage (GPU box): python tools/gemm_cold_probe.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from paella_amd import _lib

CASES = [  # name, M, N, K, mode (0 plain, 1 GRN, 2 LayerNorm), act, cfg
    ("c3 L1 mlp1 plain+bias+gelu", 32768, 5120, 1280, 0, 1, 18),
    ("c3 L1 mlp2 GRN", 32768, 1280, 5120, 1, 0, 10),
    ("c3 L1 mlp2 GRN", 32768, 1280, 5120, 1, 0, 18),
    ("c3 L1 qkv LN", 32768, 3840, 1280, 2, 0, 18),
    ("c3 L1 out plain", 32768, 1280, 1280, 0, 0, 18),
    ("b32 L1 qkv LN", 4096, 3840, 1280, 2, 0, 18),
    ("b32 L1 mlp1 plain+bias+gelu", 4096, 5120, 1280, 0, 1, 18),
]


def main():
    lib = _lib.load()
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = _lib.new_workspace(256 << 20, "cuda")
    flush = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    print("%-30s %6s %6s %6s cfg   warm us (TF)      cold us (TF)    cold/warm" % ("case", "M", "N", "K"))
    for name, M, N, K, mode, act, cfg in CASES:
        A = torch.randn(M, K, device="cuda")
        Ws = [torch.randn(N, K, device="cuda") for _ in range(3)]
        C = torch.empty(M, N, device="cuda")
        bias = torch.randn(N, device="cuda")
        rps = 256 if M % 256 == 0 else 64
        scale, shift = torch.ones(M // rps, K, device="cuda"), torch.zeros(K, device="cuda")
        stats = torch.stack([torch.zeros(M, K // 16, device="cuda"), torch.full((M, K // 16), 16.0, device="cuda")], dim=-1).contiguous()

        def run(W):
            if mode:
                rc = lib.paella_test_gemm_prologue(A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, mode, scale.data_ptr(), shift.data_ptr(), rps, stats.data_ptr(), cfg, 1,
                                                   ws.data_ptr(), ws.numel(), st())
            else:
                rc = lib.paella_op_gemm(A.data_ptr(), W.data_ptr(), bias.data_ptr() if act else None, None, C.data_ptr(), M, N, K, act, cfg, 1, ws.data_ptr(), ws.numel(), st())
            assert rc == 0, lib.paella_last_error()
        for W in Ws:
            run(W)
        torch.cuda.synchronize()
        res = {}
        for cold in (False, True):
            ts = []
            for it in range(9):
                if cold:
                    flush.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run(Ws[it % 3])
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            res[cold] = ts[len(ts) // 2]
        fl = 2.0 * M * N * K
        print("%-30s %6d %6d %6d %3d  %9.1f (%5.1f)  %9.1f (%5.1f)   %.3f" % (name, M, N, K, cfg, res[False], fl / res[False] / 1e6, res[True], fl / res[True] / 1e6, res[True] / res[False]), flush=True)
        del A, Ws, C


if __name__ == "__main__":
    main()
