"""Measure the per-kernel cost of a chain of dependent near-empty kernels (eager launches from C++), and under hipGraph."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paella_amd import _lib
lib = _lib.load()
buf = torch.zeros(1 << 20, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for blocks, n_el in ((1, 256), (256, 65536), (1024, 262144)):
    for n in (200, 2000):
        lib.paella_test_launch_chain(buf.data_ptr(), n_el, blocks, 50, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        lib.paella_test_launch_chain(buf.data_ptr(), n_el, blocks, n, st)
        e1.record()
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        print("eager  blocks=%4d n=%4d: %.2f us/kernel on GPU timeline, host enqueue %.2f us/kernel" % (blocks, n, e0.elapsed_time(e1) * 1e3 / n, th * 1e6 / n))
    # hipGraph via torch stream capture
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        sp = ctypes.c_void_p(s.cuda_stream)
        lib.paella_test_launch_chain(buf.data_ptr(), n_el, blocks, 10, sp)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            lib.paella_test_launch_chain(buf.data_ptr(), n_el, blocks, 500, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print("graph  blocks=%4d n= 500: %.2f us/kernel" % (blocks, e0.elapsed_time(e1) * 1e3 / 500))
