"""Hash of the kernel sources a libpaella_hip.so was built from.

build.py embeds it in the library (`paella_source_stamp()`); `_lib.load()` recomputes it from the sources in the tree and REFUSES a
library built from other sources -- a stale .so next to newer sources (reset mtimes, a partial copy) must not be benchmarked silently."""
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["gemm.hip", "elementwise.hip", "dwconv.hip", "attention.hip", "tail.hip", "vqgan.hip", "model.hip", "vqmodel.hip"]
HEADERS = ["common.h", "internal.h", "gemm_device.h", "philox.h", "test_hooks.h", os.path.join("..", "..", "include", "paella_hip.h")]


def source_files():
    return [os.path.normpath(os.path.join(CSRC, f)) for f in SOURCES + HEADERS]


# the compile configuration is part of what a library was "built from": changing a flag or the target must not reuse stale objects
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# per-unit extras.  attention.hip: keep the MFMA accumulators in VGPRs -- the online softmax rescales O^T between every two MFMA blocks, and with AGPR accumulators
# hipcc moved all 20 of them out and back per 16-key step (148 v_accvgpr_* per 32 keys; profiles/r05_attention_pmc_and_probe.txt)
UNIT_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def source_stamp():
    h = hashlib.sha256()
    h.update((" ".join(FLAGS) + repr(sorted(UNIT_FLAGS.items()))).encode())
    for path in source_files():
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:32]
