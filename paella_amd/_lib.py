"""ctypes binding of libpaella_hip.so (C ABI declared in include/paella_hip.h).

No torch types cross this boundary: tensors are passed as raw device pointers (tensor.data_ptr()) plus sizes,
the current torch HIP stream as a void*.  The library is built in-tree by paella_amd/build.py; there is no
fallback -- if it is missing or fails to load, every product entry point raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libpaella_hip.so")

MAX_LEVELS = 8
MAX_BLOCK_TYPES = 8
ABI_VERSION = 5


class UnetConfig(Structure):
    _fields_ = [
        ("c_in", c_int32), ("c_out", c_int32), ("num_labels", c_int32), ("c_r", c_int32), ("patch_size", c_int32),
        ("c_cond", c_int32), ("n_levels", c_int32),
        ("c_hidden", c_int32 * MAX_LEVELS), ("nhead", c_int32 * MAX_LEVELS), ("blocks", c_int32 * MAX_LEVELS),
        ("level_config", (c_char * MAX_BLOCK_TYPES) * MAX_LEVELS),
        ("clip_embd", c_int32), ("byt5_embd", c_int32), ("clip_seq_len", c_int32), ("kernel_size", c_int32),
        ("self_attn", c_int32),
    ]


class VqganConfig(Structure):
    _fields_ = [
        ("levels", c_int32), ("bottleneck_blocks", c_int32), ("c_hidden", c_int32), ("c_latent", c_int32),
        ("codebook_size", c_int32), ("scale_factor", c_float),
    ]


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
SIGNATURES = {
    "paella_abi_version": (c_int, []),
    "paella_last_error": (c_char_p, []),
    "paella_source_stamp": (c_char_p, []),
    "paella_workspace_init": (c_int, [c_void_p, c_size_t, c_void_p]),
    "paella_workspace_header_bytes": (c_size_t, []),
    "paella_unet_create": (c_int, [POINTER(UnetConfig), POINTER(c_void_p)]),
    "paella_unet_destroy": (None, [c_void_p]),
    "paella_unet_load_tensor": (c_int, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int, c_void_p]),
    "paella_unet_set_timestep_freqs": (c_int, [c_void_p, POINTER(c_float), c_int]),
    "paella_unet_finalize": (c_int, [c_void_p, c_void_p]),
    "paella_unet_cond_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "paella_unet_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int, c_int]),
    "paella_unet_cond_prepare": (c_int, [c_void_p, c_void_p, c_int, c_void_p, POINTER(c_void_p), c_int, c_int, c_void_p,
                                         c_size_t, c_void_p, c_size_t, c_void_p]),
    "paella_unet_c_embeddings": (c_int, [c_void_p, c_void_p, c_int, c_void_p, POINTER(c_void_p), c_int, c_int, c_void_p,
                                         c_void_p, c_size_t, c_void_p]),
    "paella_unet_r_embedding": (c_int, [c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p]),
    "paella_unet_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                    c_void_p, c_void_p, c_size_t, c_void_p]),
    "paella_unet_forward_shared": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_int, c_int, c_int,
                                           c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "paella_unet_forward_sample": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_int, c_int, c_int,
                                           c_void_p, c_int, c_float, c_int, c_uint64, c_void_p, c_uint64, c_int64, c_void_p, c_void_p, c_float, c_void_p,
                                           c_void_p, c_size_t, c_void_p]),
    "paella_sample_tail": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_float, c_float, c_float, c_int, c_void_p,
                                   c_uint64, c_uint64, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "paella_sample_tail_ex": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_float, c_float, c_float, c_int, c_void_p,
                                      c_uint64, c_void_p, c_uint64, c_int64, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "paella_start_tokens": (c_int, [c_uint64, c_void_p, c_int64, c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "paella_add_noise": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64, c_uint64, c_int, c_int,
                                 c_int64, c_void_p, c_void_p, c_void_p]),
    "paella_select_tokens": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "paella_vqgan_create": (c_int, [POINTER(VqganConfig), POINTER(c_void_p)]),
    "paella_vqgan_destroy": (None, [c_void_p]),
    "paella_vqgan_load_tensor": (c_int, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int, c_void_p]),
    "paella_vqgan_finalize": (c_int, [c_void_p, c_void_p]),
    "paella_vqgan_set_precision": (c_int, [c_void_p, c_int, c_void_p]),
    "paella_vqgan_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "paella_vqgan_decode_indices": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "paella_vqgan_decode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "paella_vqgan_encode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_size_t, c_void_p]),
    "paella_vqgan_quantize_rows": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "paella_vqgan_lookup_rows": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "paella_unet_set_precision": (c_int, [c_void_p, c_int, c_void_p]),
    "paella_unet_get_precision": (c_int, [c_void_p]),
    "paella_op_gemm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                               c_void_p, c_size_t, c_void_p]),
    "paella_op_layernorm": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "paella_op_dwconv_ln": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                    c_void_p]),
    "paella_op_grn_scale": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "paella_op_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                    c_int, c_int, c_void_p, c_int, c_void_p]),
}

# exported for tests / tools only; declared in paella_amd/csrc/test_hooks.h, not in the public header
TEST_HOOKS = {
    "paella_prof_detail": (c_int64, [c_void_p, c_void_p, c_int64]),
    "paella_prof_enable": (c_int, [c_int]),
    "paella_prof_collect": (c_int, [POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(c_int64)]),
    "paella_test_gemm_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "paella_test_gemm_bf16_rule": (c_int, [c_int]),
    "paella_test_grn_apply16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "paella_test_gemm_bf16_ln": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "paella_test_attention_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "paella_test_launch_chain": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "paella_test_attention_variant": (c_int, [c_int]),
    "paella_test_gemm_dma": (c_int, [c_int]),
    "paella_test_gemm_raster": (c_int, [c_int]),
    "paella_test_gemm_ring": (c_int, [c_int]),
    "paella_test_ring_resident": (ctypes.c_long, [c_int, c_int]),
    "paella_test_gemm_site": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "paella_test_gemm_site_cfg": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "paella_test_gemm_big_stagger": (c_int, [c_int]),
    "paella_test_grn_fuse": (c_int, [c_int]),
    "paella_test_ln_fold_ratio": (c_int, [c_float]),
    "paella_test_ln_guard_counter": (c_int, [c_void_p]),
    "paella_test_mlp_grn_fused": (c_int, [c_void_p] * 10 + [c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "paella_test_gemm_tail_tile": (c_int, [c_int]),
    "paella_test_tail_scores": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_float, c_float, c_float, c_uint64, c_uint64, c_int64, c_void_p, c_void_p]),
    "paella_test_gemm_prologue": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                                          c_void_p, c_size_t, c_void_p]),
}

_lib = None


class PaellaHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once). Raises PaellaHipError when it is absent -- there is no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PaellaHipError(
            "libpaella_hip.so is not built (%s). Run `python -m paella_amd.build` (needs hipcc); "
            "paella_amd has no non-HIP fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise PaellaHipError("cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, (res, args) in list(SIGNATURES.items()) + list(TEST_HOOKS.items()):
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch, let it propagate loudly
        fn.restype = res
        fn.argtypes = args
    from ._stamp import source_stamp
    try:
        here = source_stamp()
    except OSError as e:  # the kernel sources ship next to the library; without them its provenance cannot be checked
        raise PaellaHipError("cannot verify %s: the kernel sources it is checked against are missing (%s)" % (LIB_PATH, e)) from e
    built = lib.paella_source_stamp().decode()
    if built != here:
        raise PaellaHipError("libpaella_hip.so was built from other sources (library stamp %s, sources in the tree %s): rebuild with "
                             "`python -m paella_amd.build`; a stale library is never used silently" % (built, here))
    if lib.paella_abi_version() != ABI_VERSION:
        raise PaellaHipError("libpaella_hip.so ABI %d != binding ABI %d" % (lib.paella_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().paella_last_error()
        raise PaellaHipError("libpaella_hip error %d: %s" % (rc, msg.decode("utf-8", "replace") if msg else "?"))


def ptr(t):
    """Device/host pointer of a tensor (or None)."""
    return None if t is None else c_void_p(t.data_ptr())


def new_workspace(nbytes, device):
    """A caller-owned workspace with its ticket header initialised (include/paella_hip.h: paella_workspace_init)."""
    import torch
    lib = load()
    nbytes = max(int(nbytes), int(lib.paella_workspace_header_bytes()))
    with torch.cuda.device(device):
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        check(lib.paella_workspace_init(ptr(ws), ws.numel(), stream_ptr(device)))
    return ws


def stream_ptr(device=None):
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)
