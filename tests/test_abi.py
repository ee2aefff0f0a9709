"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol include/paella_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "paella_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(paella_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built_lib):
    names = header_functions()
    assert len(names) >= 25
    raw = ctypes.CDLL(os.path.join(ROOT, "paella_amd", "csrc", "libpaella_hip.so"))
    for n in names:
        assert hasattr(raw, n), "libpaella_hip.so does not export " + n


def test_binding_table_matches_header(built_lib):
    from paella_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_functions()
    assert built_lib.paella_abi_version() == _lib.ABI_VERSION


def test_no_torch_in_abi():
    src = open(os.path.join(ROOT, "include", "paella_hip.h")).read()
    assert "torch" not in re.sub(r"/\*.*?\*/", "", src, flags=re.S).lower()
    assert "at::" not in src and "Tensor" not in re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def test_argument_validation_without_gpu(built_lib):
    """create() validates configurations on the host (no device work)."""
    from paella_amd import _lib
    c = _lib.UnetConfig()
    h = ctypes.c_void_p()
    c.n_levels = 0
    assert built_lib.paella_unet_create(ctypes.byref(c), ctypes.byref(h)) == -1
    assert b"n_levels" in built_lib.paella_last_error()
    v = _lib.VqganConfig()
    v.levels, v.c_hidden, v.c_latent, v.codebook_size, v.bottleneck_blocks = 9, 384, 4, 8192, 12
    assert built_lib.paella_vqgan_create(ctypes.byref(v), ctypes.byref(h)) == -1


def test_product_never_imports_oracle():
    """The product package must not route through the oracle or any CPU fallback."""
    pkg = os.path.join(ROOT, "paella_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f


def test_stale_library_is_refused(built_lib, monkeypatch):
    """The library carries the hash of the sources it was built from (paella_source_stamp); the binding recomputes it from the tree and refuses a
    library built from other sources instead of benchmarking it silently (build.py trusted modification times)."""
    from paella_amd import _lib, _stamp, build
    assert built_lib.paella_source_stamp().decode() == _stamp.source_stamp() == build.built_stamp()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_stamp, "source_stamp", lambda: "0" * 32)
    with pytest.raises(_lib.PaellaHipError, match="built from other sources"):
        _lib.load()


def test_test_hooks_header_binding_and_library_agree(built_lib):
    """paella_amd/csrc/test_hooks.h (tools / tests only, not the public ABI) is included by the translation units that define the hooks, so the compiler checks the
    prototypes; here: the header, the ctypes table and the exported symbols name the same set, and none of them leaks into the public header."""
    src = open(os.path.join(ROOT, "paella_amd", "csrc", "test_hooks.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    declared = sorted(set(re.findall(r"\b(paella_[a-z0-9_]+)\s*\(", src)))
    from paella_amd import _lib
    assert declared == sorted(_lib.TEST_HOOKS), (set(declared) ^ set(_lib.TEST_HOOKS))
    raw = ctypes.CDLL(os.path.join(ROOT, "paella_amd", "csrc", "libpaella_hip.so"))
    for n in declared:
        assert hasattr(raw, n), "libpaella_hip.so does not export " + n
    assert not set(declared) & set(header_functions())


def test_library_reads_no_environment_variables():
    """A/B switches are test hooks; a deployment's kernels must not depend on the environment (three getenv knobs lived in gemm.hip)."""
    csrc = os.path.join(ROOT, "paella_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f
