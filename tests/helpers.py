"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np
import torch

from oracle import golden_configs as G
from paella_amd import synth


def weights_for(module, cfg_blocks, golden_npz=None, seed=G.WEIGHT_SEED):
    """Seeded synthetic weights for `module`; verifies key/shape compatibility and the checksum stored with the fixture."""
    sd = synth.synth_state_dict(module.state_dict(), seed=seed, n_blocks=cfg_blocks)
    if golden_npz is not None:
        keys = sorted(k + ":" + ",".join(str(d) for d in v.shape) for k, v in module.state_dict().items())
        assert keys == sorted(golden_npz["keys"].tolist()), "state-dict keys/shapes differ from the reference's"
        c = synth.checksum(sd)
        np.testing.assert_allclose(np.array(c), golden_npz["checksum"], rtol=1e-9, err_msg="synthetic weight generator drifted")
    module.load_state_dict(sd)
    return sd


def cond_for(cfg, B, S_byt5, n_img, seed, device="cpu"):
    return synth.synth_conditioning(B, S_byt5, cfg["byt5_embd"], cfg["clip_embd"], seed=seed, with_clip=True, n_clip_image=n_img,
                                    device=device)


def to_dev(inputs, device):
    out = {}
    for k, v in inputs.items():
        if v is None:
            out[k] = None
        elif isinstance(v, (list, tuple)):
            out[k] = [t.to(device) for t in v]
        else:
            out[k] = v.to(device)
    return out


def argmax_report(ref_logits, got_logits, eps=1e-4):
    """Near-tie policy: positions whose reference top1-top2 margin is below eps are counted
    separately, never silently dropped.  logits [B, L, H, W].  Returns (mismatches_clear, mismatches_near_tie, n_near_tie)."""
    top2 = ref_logits.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    mism = ref_logits.argmax(1) != got_logits.argmax(1)
    near = margin < eps
    return int((mism & ~near).sum()), int((mism & near).sum()), int(near.sum())
