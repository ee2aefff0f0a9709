"""Where do the large fp32 GEMMs lose their last ~20 %?  Phase ablation of the real main loop (timing only: results are wrong).
Builds probe copies of the library from a PATCHED COPY of paella_amd/csrc/gemm.hip (the product source is not touched) in which the
unit step skips  1: the global operand loads   2: the LDS tile stores   4: the workgroup barrier   8: the fragment reads (PIPE tiles)
and times one large GEMM with each.
    python tools/probes/gemm_ablate.py --build          (anywhere; needs the normal build's objects in paella_amd/csrc/build)
    python tools/probes/gemm_ablate.py                  (GPU box)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "paella_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "probes")
VARIANTS = [0, 1, 2, 4, 8, 3, 6, 7, 15]
sys.path.insert(0, ROOT)


def lib_path(v):
    return os.path.join(OUT, "libpaella_hip_abl%d.so" % v)


def build():
    from paella_amd import build as B
    B.build()
    src = open(os.path.join(CSRC, "gemm.hip")).read()
    edits = [("        fetch(rf, sl ^ 1);  // DMA operands", "        if (!(ABL & 1)) fetch(rf, sl ^ 1);  // DMA operands"),
             ("            read_group(I1{}, sl);  // k group 1", "            if (!(ABL & 8)) read_group(I1{}, sl);  // k group 1"),
             ("            store_unit(rs, sl ^ 1);\n            __syncthreads();\n            read_group(I0{}, sl ^ 1);", "            if (!(ABL & 2)) store_unit(rs, sl ^ 1);\n            if (!(ABL & 4)) __syncthreads();\n            if (!(ABL & 8)) read_group(I0{}, sl ^ 1);"),
             ("            compute(sl);\n            __builtin_amdgcn_sched_barrier(0);\n            store_unit(rs, sl ^ 1);\n            __syncthreads();", "            compute(sl);\n            __builtin_amdgcn_sched_barrier(0);\n            if (!(ABL & 2)) store_unit(rs, sl ^ 1);\n            if (!(ABL & 4)) __syncthreads();")]
    for a, b in edits:
        assert src.count(a) == 1, "gemm.hip changed: the ablation patch no longer applies (%r)" % a[:40]
        src = src.replace(a, b)
    tmpdir = os.path.join(OUT, "build")
    os.makedirs(tmpdir, exist_ok=True)
    objs = [os.path.join(CSRC, "build", s.replace(".hip", ".o")) for s in B.SOURCES if s != "gemm.hip"]
    procs = []
    for v in VARIANTS:
        s = os.path.join(tmpdir, "gemm_abl%d.hip" % v)
        open(s, "w").write("#define ABL %d\n" % v + src)
        o = s.replace(".hip", ".o")
        procs.append((v, o, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", "-I", CSRC, "-c", s, "-o", o])))
        if len(procs) % 3 == 0:
            for _, _, p in procs[-3:]:
                assert p.wait() == 0
    for v, o, p in procs:
        assert p.wait() == 0
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path(v), o] + objs)
        print("built", lib_path(v))


def main():
    if "--build" in sys.argv:
        return build()
    import torch
    vp = ctypes.c_void_p
    names = {1: "no global loads", 2: "no LDS stores", 4: "no barrier", 8: "no fragment reads (PIPE tiles only)"}
    shapes = [("32768x5120x1280", 32768, 5120, 1280)]
    for name, M, N, K in shapes:
        A = torch.randn(M, K, device="cuda")
        Ws = [torch.randn(N, K, device="cuda") for _ in range(3)]
        C = torch.empty(M, N, device="cuda")
        ws = torch.zeros(256 << 20, dtype=torch.uint8, device="cuda")
        st = lambda: vp(torch.cuda.current_stream().cuda_stream)
        print("%s, fp32, one tile per workgroup; us per launch (TFLOP/s equivalent) by skipped phases" % name)
        for cfg, label in [(10, "128x128 8 waves, PIPE loop"), (18, "64x64 4 waves, 1-deep ring")]:
            row = []
            for v in VARIANTS:
                lib = ctypes.CDLL(lib_path(v))
                lib.paella_op_gemm.argtypes = [vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, vp]
                run = lambda W: lib.paella_op_gemm(vp(A.data_ptr()), vp(W.data_ptr()), None, None, vp(C.data_ptr()), M, N, K, 0, cfg, 1, vp(ws.data_ptr()), ws.numel(), st())
                for W in Ws:
                    assert run(W) == 0
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(12):
                    run(Ws[i % 3])
                e1.record()
                e1.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 12
                what = "full kernel" if v == 0 else " + ".join(n for b, n in names.items() if v & b)
                row.append("    %-62s %8.1f us  (%5.1f TF)" % (what, us, 2.0 * M * N * K / us / 1e6))
            print("  tile config %d (%s):" % (cfg, label))
            print("\n".join(row))


if __name__ == "__main__":
    main()
