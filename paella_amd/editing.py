"""Inpainting / structural editing on the token grid (BASELINE config 5).

The reference snapshot has no dedicated function for this (it lived in the missing notebook); it is the composition of
on-disk pieces, reproduced here with the same calls:
    VQModel.encode(img)[2]                                   -> tokens              (src/vqgan.py:91-95)
    Paella.add_noise(tokens, t, mask=user_mask, random_x=..) -> masked renoise      (src/modules.py:277-283)
    sample(..., init_x=noised, t_start<1)                    -> denoise             (src_distributed/utils.py:97-109)
    VQModel.decode_indices(tokens)                           -> image               (src/vqgan.py:103-107)
EXTENSION (not reference behaviour, labelled as such): `keep_known=True` re-imposes the known tokens on the result.
"""
import torch

from .sampling import sample_distributed


def inpaint(model, vqgan, images, mask, model_inputs, unconditional_inputs, steps=12, t_start=1.0, temperature=(0.7, 0.3),
            cfg=(8.0, 8.0), keep_known=True, decode=True, random_x=None, **kwargs):
    """images fp32 [B,3,Hp,Wp] in [0,1]; mask int/bool [B,h,w] on the TOKEN grid (1 = regenerate); random_x (optional) the tokens
    add_noise writes into the masked region (default: torch.randint_like, as Paella.add_noise draws them).
    Returns (tokens, image or None)."""
    tokens = vqgan.encode(images)[2]
    mask = mask.to(device=tokens.device, dtype=torch.int64)
    if mask.shape != tokens.shape:
        raise ValueError("mask must be given on the token grid %s" % (tuple(tokens.shape),))
    B = tokens.size(0)
    t = torch.full((B,), float(t_start), device=tokens.device)
    noised, _ = model.add_noise(tokens, t, mask=mask, random_x=random_x)
    out = sample_distributed(model, model_inputs, unconditional_inputs, tuple(tokens.shape), init_x=noised, steps=steps,
                             temperature=temperature, cfg=cfg, t_start=t_start, **kwargs)
    if keep_known:  # extension: the reference's sample() may also rewrite known positions
        out = out * mask + tokens * (1 - mask)
    return out, (vqgan.decode_indices(out) if decode else None)
