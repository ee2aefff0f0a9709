"""Build libpaella_hip.so (the gfx950 kernels + C ABI) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the authoring container and on the GPU box.
The shared library lands next to the sources (paella_amd/csrc/libpaella_hip.so) so that it travels with the
repository snapshot and shows up as an in-tree native library when loaded.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["gemm.hip", "gemm_bf16.hip", "elementwise.hip", "dwconv.hip", "attention.hip", "tail.hip", "vqgan.hip", "model.hip", "vqmodel.hip"]
HEADERS = ["common.h", "internal.h", "gemm_device.h", "philox.h", "test_hooks.h", os.path.join("..", "..", "include", "paella_hip.h")]
LIB = os.path.join(CSRC, "libpaella_hip.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the Paella HIP library needs ROCm's hipcc")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link the shared library. Returns its path."""
    headers = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    if not force and not _stale(LIB, [os.path.join(CSRC, s) for s in SOURCES] + headers):
        return LIB  # prebuilt library (e.g. shipped to the GPU box) is newer than every source
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (s, r.stdout, r.stderr))
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(objdir, src.replace(".hip", ".o")) for src in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print("built", path)
