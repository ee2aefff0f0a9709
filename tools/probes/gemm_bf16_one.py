"""One bf16 GEMM launch configuration, repeated: the profiled command of tools/probes/pp_counters.sh (rocprofv3 --pmc ... -- python tools/probes/gemm_bf16_one.py M N K tile splitk [reps] [store]).
store = 0: no output at all (main loop + ramp only), 1 (default): fp32 output."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from paella_amd import _lib

M, N, K, tile, sk = (int(v) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 6
store = int(sys.argv[7]) if len(sys.argv) > 7 else 1
lib = _lib.load()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = _lib.new_workspace(256 << 20, "cuda")
A = torch.randn(M, K, device="cuda").bfloat16()
Ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16() for _ in range(3)]
C = torch.empty(M, N, device="cuda") if store else None
ts = []
for i in range(reps):
    W = Ws[i % 3]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = lib.paella_test_gemm_bf16(A.data_ptr(), W.data_ptr(), None, None, C.data_ptr() if store else None, None, M, N, K, 0, None, tile, sk, ws.data_ptr(), ws.numel(), st)
    e1.record()
    assert rc == 0, lib.paella_last_error()
    e1.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts = sorted(ts[1:])
us = ts[len(ts) // 2]
bm, bn = {36: (256, 128), 37: (256, 256), 18: (64, 64)}.get(tile, (0, 0))
tiles = (-(-M // bm)) * (-(-N // bn)) if bm else 0
print("%7d x %7d x %5d  tile %2d/%-5d store %d : %9.1f us  %7.1f TFLOP/s   %5d tiles, %6.2f us per tile and CU-round (%d rounds of 256)" %
      (M, N, K, tile, sk, store, us, 2.0 * M * N * K / us / 1e6, tiles, us / max(1, -(-tiles // 256)), -(-tiles // 256)))
