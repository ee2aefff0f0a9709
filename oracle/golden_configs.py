"""Model / input configurations shared by oracle/make_golden.py and the tests (test infrastructure)."""

UNET_TINY = dict(c_in=32, c_out=32, num_labels=64, c_r=16, patch_size=2, c_cond=64, c_hidden=[32, 64, 64], nhead=[-1, 4, 4],
                 blocks=[1, 2, 1], level_config=['CT', 'CTA', 'CTA'], clip_embd=48, byt5_embd=40, clip_seq_len=4, kernel_size=3,
                 dropout=0.1, self_attn=True)

# head_dim 80 like the released model, 3 levels, ragged conditioning
UNET_MID = dict(c_in=64, c_out=64, num_labels=1024, c_r=64, patch_size=2, c_cond=128, c_hidden=[160, 320, 320], nhead=[-1, 4, 4],
                blocks=[2, 3, 2], level_config=['CT', 'CTA', 'CTA'], clip_embd=96, byt5_embd=72, clip_seq_len=4, kernel_size=3,
                dropout=0.1, self_attn=True)

# exercises block type F, un-fusable TimestepBlock positions, cross-attention only, patch_size 1, two levels
UNET_VARIANT = dict(c_in=32, c_out=32, num_labels=128, c_r=32, patch_size=1, c_cond=64, c_hidden=[32, 64], nhead=[-1, 2],
                    blocks=[1, 2], level_config=['CFT', 'TAC'], clip_embd=48, byt5_embd=40, clip_seq_len=2, kernel_size=3,
                    dropout=0.0, self_attn=False)

VQ_TINY_F4 = dict(levels=2, bottleneck_blocks=2, c_hidden=32, c_latent=4, codebook_size=64, scale_factor=0.3764)
VQ_TINY_F8 = dict(levels=3, bottleneck_blocks=2, c_hidden=64, c_latent=4, codebook_size=128, scale_factor=0.3764)

# the "573M-class" stand-in of BASELINE configs 2/3 and the released 1B default
UNET_570M = dict(c_in=256, c_out=256, num_labels=8192, c_r=64, patch_size=2, c_cond=1024, c_hidden=[640, 1280, 1280],
                 nhead=[-1, 16, 16], blocks=[4, 8, 4], level_config=['CT', 'CTA', 'CTA'], clip_embd=1024, byt5_embd=1536,
                 clip_seq_len=4, kernel_size=3, dropout=0.1, self_attn=True)
UNET_1B = dict(UNET_570M, blocks=[6, 16, 6])
VQ_F8 = dict(levels=3, bottleneck_blocks=12, c_hidden=384, c_latent=4, codebook_size=8192, scale_factor=0.3764)

WEIGHT_SEED = 0
COND_SEED = 2
SAMPLER_SEED = 42


import torch


def train_step_inputs(cfg):
    """Seeded inputs of one training step on the tiny config (shared with tests/test_training.py through the fixture)."""
    B, H, W = 2, 16, 16
    g = torch.Generator().manual_seed(21)
    latents = torch.randint(0, cfg["num_labels"], (B, H, W), generator=g)
    t = (1 - torch.rand(B, generator=g)).add(0.001).clamp(0.001, 1.0)
    mask = (torch.rand(B, H, W, generator=g) <= t[:, None, None]).long()
    random_x = torch.randint(0, cfg["num_labels"], (B, H, W), generator=g)
    from paella_amd import synth
    c = synth.synth_conditioning(B, 5, cfg["byt5_embd"], cfg["clip_embd"], seed=COND_SEED + 9, with_clip=True, n_clip_image=1)
    return latents, t, mask, random_x, c


