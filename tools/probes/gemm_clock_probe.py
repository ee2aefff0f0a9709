"""Which shader clock do the real fp32 GEMM launches run at?  Builds a probe copy of the library (-DPAELLA_GEMM_CLOCK_PROBE: one
workgroup per launch stamps s_memtime shader cycles and the 100 MHz wall clock at entry and exit) and times large GEMMs with it.
The matrix-core peak that a launch can reach is 64 flop x 1024 SIMDs x that clock (tools/probes/mfma_peak_probe.hip shows
155.6 TFLOP/s = 2.39 GHz for MFMA-only code with constant operands).
Build (anywhere, cross-compiles):  python tools/probes/gemm_clock_probe.py --build
Run (GPU box):                     python tools/probes/gemm_clock_probe.py"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "paella_amd", "csrc")
LIB = os.path.join(ROOT, "tools", "probes", "libpaella_hip_clockprobe.so")
sys.path.insert(0, ROOT)


def build():
    from paella_amd import build as B
    srcs = [os.path.join(CSRC, s) for s in B.SOURCES]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", "-shared", "-DPAELLA_GEMM_CLOCK_PROBE", "-o", LIB] + srcs
    print(" ".join(cmd))
    subprocess.check_call(cmd)


def main():
    if "--build" in sys.argv:
        return build()
    import torch
    lib = ctypes.CDLL(LIB)
    lib.paella_last_error.restype = ctypes.c_char_p
    vp = ctypes.c_void_p
    lib.paella_op_gemm.argtypes = [vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, vp]
    lib.paella_workspace_init.argtypes = [vp, ctypes.c_size_t, vp]
    ws = torch.zeros(256 << 20, dtype=torch.uint8, device="cuda")
    st = lambda: vp(torch.cuda.current_stream().cuda_stream)
    assert lib.paella_workspace_init(vp(ws.data_ptr()), ws.numel(), st()) == 0
    shapes = [("c3 L1 mlp1", 32768, 5120, 1280, 10), ("c3 L1 mlp1", 32768, 5120, 1280, 18), ("c3 L0 mlp1", 131072, 2560, 640, 10),
              ("b32 L1 mlp1", 4096, 5120, 1280, 10), ("b1 L1 mlp1", 128, 5120, 1280, -1)]
    for name, M, N, K, cfg in shapes:
        A = torch.randn(M, K, device="cuda")
        Ws = [torch.randn(N, K, device="cuda") for _ in range(3)]
        C = torch.empty(M, N, device="cuda")
        run = lambda W: lib.paella_op_gemm(vp(A.data_ptr()), vp(W.data_ptr()), None, None, vp(C.data_ptr()), M, N, K, 0, cfg, 1, vp(ws.data_ptr()), ws.numel(), st())
        for W in Ws:
            assert run(W) == 0, lib.paella_last_error()
        torch.cuda.synchronize()
        reps = 20 if M >= 4096 else 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            run(Ws[i % 3])
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        out = (ctypes.c_ulonglong * 2)()
        assert lib.paella_probe_gemm_clock(out) == 0
        ghz = out[0] / (out[1] / 100e6) / 1e9 if out[1] else float("nan")
        tf = 2.0 * M * N * K / us / 1e6
        print("%-12s %7dx%5dx%5d tile cfg %3d: %9.1f us/launch  %6.1f TFLOP/s | shader clock during the launch %.3f GHz -> matrix-core peak at that clock %.1f TFLOP/s, "
              "this launch = %.1f %% of it" % (name, M, N, K, cfg, us, tf, ghz, 65.536 * ghz, 100.0 * tf / (65.536 * ghz)))


if __name__ == "__main__":
    main()
