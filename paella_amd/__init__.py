"""paella_amd -- the sampling hot path of dome272/Paella, built MI355X-first.

Public surface (mirrors the reference's Python API):
    Paella / DenoiseUNet      reference src/modules.py:109
    VQModel                   reference src/vqgan.py:45
    sample                    reference src/utils.py:35
    sample_distributed        reference src_distributed/utils.py:97
    replace_attention_layers  reference utils/alter_attention.py:45
    load_conditional_models   reference src_distributed/utils.py:65 (+ embed_prompts, load_checkpoint: paella_amd/conditioning.py)
Everything executes through libpaella_hip.so (hand-written HIP for gfx950, C ABI in include/paella_hip.h).
The opt-in bf16 fast mode is a per-model switch: `Paella.set_gemm_precision("bf16")` (outside the fp32 parity contract).
"""
from .conditioning import build_paella, embed_prompts, load_checkpoint, load_conditional_models
from .editing import GraphInpainter, inpaint
from .modules import CondCache, DenoiseUNet, Paella, replace_attention_layers
from .sampling import GraphSampler, sample, sample_distributed, select_tokens
from .vqgan import VectorQuantize, VQModel


__all__ = ["Paella", "DenoiseUNet", "CondCache", "VQModel", "VectorQuantize", "sample", "sample_distributed", "GraphSampler",
           "replace_attention_layers", "inpaint", "GraphInpainter", "select_tokens", "load_conditional_models", "embed_prompts", "load_checkpoint", "build_paella"]
