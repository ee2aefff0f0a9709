"""Does a large fp32 GEMM keep its isolated rate under sustained load?  Launches one shape back to back for several seconds and prints the
rate per window, next to the shader clock / power the SMI reports -- in-model GEMMs run ~10 % below their isolated sweeps (profiles/r03_gemm_by_shape_*).
Usage (GPU box): python tools/sustained_probe.py [--cfg 18] [--seconds 6] [--vendor]"""
import argparse
import ctypes
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from paella_amd import _lib


def smi_sampler(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5)
            out.append((time.perf_counter(), r.stdout.strip().replace("\n", " | ")))
        except Exception as e:  # the probe must not die on a missing SMI
            out.append((time.perf_counter(), "smi failed: %r" % (e,)))
        stop.wait(0.5)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=18)
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--shape", default="32768,5120,1280")
    ap.add_argument("--vendor", action="store_true", help="torch.mm (vendor BLAS) instead of the library kernel")
    ap.add_argument("--act", type=int, default=0)
    ap.add_argument("--idle", type=float, default=0.0, help="seconds of idle before the run (cool start)")
    a = ap.parse_args()
    M, N, K = (int(v) for v in a.shape.split(","))
    lib = _lib.load()
    A = torch.randn(M, K, device="cuda")
    Ws = [torch.randn(N, K, device="cuda") for _ in range(3)]
    C = torch.empty(M, N, device="cuda")
    bias = torch.randn(N, device="cuda")
    ws = _lib.new_workspace(256 << 20, "cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(i):
        W = Ws[i % 3]
        if a.vendor:
            torch.mm(A, W.t(), out=C)
        else:
            rc = lib.paella_op_gemm(A.data_ptr(), W.data_ptr(), bias.data_ptr() if a.act else None, None, C.data_ptr(), M, N, K, a.act, a.cfg, 1, ws.data_ptr(), ws.numel(), st)
            assert rc == 0, lib.paella_last_error()
    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    if a.idle:
        time.sleep(a.idle)
    stop, smi = threading.Event(), []
    th = threading.Thread(target=smi_sampler, args=(stop, smi))
    th.start()
    t_start = time.perf_counter()
    win = 20
    rows = []
    i = 0
    while time.perf_counter() - t_start < a.seconds:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(win):
            run(i)
            i += 1
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / win
        rows.append((time.perf_counter() - t_start, us, 2.0 * M * N * K / us / 1e6))
    stop.set()
    th.join()
    print("# %s %dx%dx%d cfg %s act %d: window of %d launches" % ("vendor torch.mm" if a.vendor else "paella_op_gemm", M, N, K, a.cfg, a.act, win))
    for t, us, tf in rows[:: max(1, len(rows) // 24)]:
        print("t=%5.2fs  %8.1f us  %6.1f TFLOP/s" % (t, us, tf))
    print("first window %.1f TF, last window %.1f TF, min %.1f, max %.1f" % (rows[0][2], rows[-1][2], min(r[2] for r in rows), max(r[2] for r in rows)))
    for t, line in smi[:: max(1, len(smi) // 8)]:
        print("smi t=%5.2fs %s" % (t - t_start, line[:400]))


if __name__ == "__main__":
    main()
