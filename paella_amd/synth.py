"""Seeded synthetic weights and conditioning for tests and benchmarks (no checkpoints are available offline).

The reference's own initialisation makes the logits identically zero (clf weight = 0, TimestepBlock weights = 0,
GRN gamma/beta = 0, VQGAN gammas = 0 -- SURVEY D7), which would make every parity test vacuous.  `synth_state_dict`
draws every tensor of a state dict from a CPU generator, by key, with scales that keep activations O(1) through
the stack so logits have a meaningful spread (std ~ 1).  CPU generation is deterministic for a given torch
version, so the GPU box regenerates exactly the tensors the golden fixtures were produced with; the fixtures also
store a checksum of the weights to catch any drift.
"""
import math

import torch


def _std_for(key, shape, n_blocks):
    if key.endswith("gammas"):            # VQGAN ResBlock scalars
        return 0.5
    if key.endswith("gamma") or key.endswith("beta"):   # GRN
        return 0.3
    if key.endswith("running_mean"):
        return 0.3
    if key.endswith("bias"):
        return 0.05
    if key == "vquantizer.codebook.weight":
        return 1.0
    if key == "in_mapper.0.weight":
        return 1.0
    if key.endswith("mapper.weight") and len(shape) == 2 and "kv_mapper" not in key and not key.startswith(("byt5", "clip")):
        return 0.5 / math.sqrt(shape[1])  # TimestepBlock mapper
    if len(shape) >= 2:
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        if key.endswith("depthwise.weight") or key.endswith("depthwise.1.weight"):
            fan_in = shape[1] * shape[2] * shape[3]
        std = 1.0 / math.sqrt(fan_in)
        # residual-branch outputs are damped so the trunk stays O(1) over n_blocks blocks
        if key.endswith("channelwise.4.weight") or key.endswith("channelwise.2.weight") or key.endswith("out_proj.weight"):
            std /= math.sqrt(max(n_blocks, 1)) * 0.5
        return std
    return 0.05


def synth_state_dict(reference_sd, seed=0, n_blocks=8):
    """Return {key: fp32 CPU tensor} with the shapes of `reference_sd`, drawn deterministically from `seed`."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = {}
    for key in sorted(reference_sd.keys()):
        t = reference_sd[key]
        shape = tuple(t.shape)
        if key.endswith("num_batches_tracked"):
            out[key] = torch.tensor(0, dtype=torch.long)
            continue
        if key.endswith("running_var"):
            out[key] = torch.rand(shape, generator=g) + 0.5
            continue
        v = torch.randn(shape, generator=g) * _std_for(key, shape, n_blocks)
        if key.endswith(".1.weight") and len(shape) == 1:   # BatchNorm weight
            v = v + 1.0
        out[key] = v.float()
    return out


def checksum(sd):
    """Order-independent fingerprint of a state dict (float64 sums), stored next to golden outputs."""
    s1 = s2 = 0.0
    for key in sorted(sd.keys()):
        t = sd[key].double()
        s1 += float(t.sum())
        s2 += float((t * t).sum())
    return s1, s2


def randomize_(module, seed=0, n_blocks=None):
    """Load seeded synthetic weights into a paella_amd (or reference) module, in place, on its current device."""
    sd = module.state_dict()
    if n_blocks is None:
        cfg = getattr(module, "_cfg", {})
        n_blocks = sum(cfg.get("blocks", [4])) if "blocks" in cfg else cfg.get("bottleneck_blocks", 8)
    new = synth_state_dict(sd, seed=seed, n_blocks=n_blocks)
    module.load_state_dict(new)
    return new


def synth_conditioning(B, S_byt5, byt5_embd, clip_embd, seed=2, with_clip=True, n_clip_image=0, device="cpu"):
    """Random text / image embeddings of the shapes Paella.forward expects."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = {"byt5": torch.randn(B, S_byt5, byt5_embd, generator=g).to(device),
           "clip": torch.randn(B, clip_embd, generator=g).to(device) if with_clip else None}
    if n_clip_image == 0:
        out["clip_image"] = None
    elif n_clip_image == 1:
        out["clip_image"] = torch.randn(B, clip_embd, generator=g).to(device)
    else:
        out["clip_image"] = [torch.randn(B, clip_embd, generator=g).to(device) for _ in range(n_clip_image)]
    return out
