"""paella_amd -- the sampling hot path of dome272/Paella, built MI355X-first.

Public surface (mirrors the reference's Python API):
    Paella / DenoiseUNet      reference src/modules.py:109
    VQModel                   reference src/vqgan.py:45
    sample                    reference src/utils.py:35
    sample_distributed        reference src_distributed/utils.py:97
    replace_attention_layers  reference utils/alter_attention.py:45
    load_conditional_models   reference src_distributed/utils.py:65 (+ embed_prompts, load_checkpoint: paella_amd/conditioning.py)
Everything executes through libpaella_hip.so (hand-written HIP for gfx950, C ABI in include/paella_hip.h).
"""
from .conditioning import build_paella, embed_prompts, load_checkpoint, load_conditional_models
from .editing import inpaint
from .modules import CondCache, DenoiseUNet, Paella, replace_attention_layers
from .sampling import GraphSampler, sample, sample_distributed
from .vqgan import VectorQuantize, VQModel


def set_gemm_precision(mode):
    """OPT-IN fast mode, outside the fp32 parity contract: "bf16" sends every dense contraction through bf16-operand MFMA
    with fp32 accumulation (weights from a bf16 shadow copy, activations rounded on the way into the matrix cores);
    "fp32" (default) is the exact path.  Process-wide."""
    from . import _lib
    modes = {"fp32": 0, "f32": 0, "bf16": 1}
    if mode not in modes:
        raise ValueError("gemm precision must be 'fp32' or 'bf16'")
    _lib.check(_lib.load().paella_set_gemm_precision(modes[mode]))


def get_gemm_precision():
    from . import _lib
    return "bf16" if _lib.load().paella_get_gemm_precision() == 1 else "fp32"

__all__ = ["Paella", "DenoiseUNet", "CondCache", "VQModel", "VectorQuantize", "sample", "sample_distributed", "GraphSampler", "set_gemm_precision", "get_gemm_precision",
           "replace_attention_layers", "inpaint", "load_conditional_models", "embed_prompts", "load_checkpoint", "build_paella"]
