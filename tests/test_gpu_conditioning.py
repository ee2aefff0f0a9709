"""GPU: prompts -> (config-initialised) ByT5 / CLIP encoders -> checkpoint-loaded Paella + VQGAN -> sample() -> decode, end to end
(reference src_distributed/utils.py:65-82 + train.py:143-166).  Encoder WEIGHTS are random (none exist offline): the test
pins the plumbing -- the reference's checkpoint file layout feeds the HIP engine, the front-end's embeddings are what the sampler consumes,
and with those embeddings the HIP sampler reproduces the CPU oracle token for token."""
import os

import numpy as np
import pytest
import torch

import paella_amd
from oracle import golden_configs as G
from oracle import paella_oracle as O
from paella_amd import conditioning as C
from paella_amd import synth
from tests.helpers import assert_token_parity, stepwise_token_parity
from tests.test_conditioning import small_configs

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_prompts_to_image_through_checkpoint_files(tmp_path, built_lib):
    cfg, vcfg = G.UNET_TINY, G.VQ_TINY_F8
    # --- "released" checkpoints in the reference's layout ---
    src = paella_amd.Paella(**cfg)
    sd = synth.randomize_(src, seed=0)
    vq_src = paella_amd.VQModel(**vcfg)
    vsd = synth.randomize_(vq_src, seed=0)
    unet_path, vq_path = os.path.join(tmp_path, "paella.pt"), os.path.join(tmp_path, "vqgan.pt")
    torch.save({"state_dict": src.state_dict(), "iter": 1}, unet_path)
    torch.save({"state_dict": vq_src.state_dict()}, vq_path)

    byt5_cfg, clip_cfg = small_configs()
    torch.manual_seed(0)
    vqgan, clip_triple, byt5_pair = C.load_conditional_models(("ViT-H-14", "laion2b_s32b_b79k"), "google/byt5-xl", vq_path, DEV, encoders="config",
                                                              vqgan_kwargs=vcfg, byt5_config=byt5_cfg, clip_config=clip_cfg)
    model = C.build_paella(unet_path, device=DEV, **cfg)
    assert next(byt5_pair[1].parameters()).is_cuda and next(clip_triple[1].parameters()).is_cuda

    captions = ["a pan of paella on a wooden table", "fog"]
    images = torch.rand(2, 3, 256, 256, device=DEV)
    cond, uncond = C.embed_prompts(captions, byt5_pair, clip_triple, images=images)
    assert cond["byt5"].is_cuda and cond["byt5"].shape[0] == 2 and cond["clip_image"].shape == (2, cfg["clip_embd"])

    # --- the HIP sampler on the front-end's embeddings vs the CPU oracle on the same embeddings and noise: token for token ---
    steps = 4
    noise = O.replay_torch_noise(G.SAMPLER_SEED, (2, 16, 16), cfg["num_labels"], steps, steps - 1)
    cpu = lambda d: {k: (None if v is None else v.cpu()) for k, v in d.items()}

    def fwd(tk, rr, **inp):
        with torch.no_grad():
            return O.unet_forward(sd, cfg, tk, rr, **inp)
    rep = stepwise_token_parity(model, fwd, cfg["num_labels"], cpu(cond), cpu(uncond), cond, uncond, noise, steps, steps - 1, (1.0, 0.2), 8.0)
    assert_token_parity(rep, "front-end embeddings -> HIP sampler vs oracle, teacher-forced")
    toks = paella_amd.sample(model, cond, (2, 16, 16), unconditional_inputs=uncond, steps=steps, renoise_steps=steps - 1, device=DEV, noise=noise)
    t_list = [float(v) for v in torch.linspace(1.0, 0.0, steps + 1)]
    temps = [float(v) for v in torch.linspace(1.0, 0.2, steps)]
    cf = (float(torch.tensor(8.0)), float(torch.tensor(1.0 - 8.0)))
    with torch.no_grad():
        otoks, _ = O.sample(fwd, cfg["num_labels"], cpu(cond), cpu(uncond), (2, 16, 16), steps=steps, renoise_steps=steps - 1, temperatures=temps,
                            cfgs=[cf] * steps, t_list=t_list, noise=noise)
    n_diff = int((toks.cpu() != otoks).sum())
    print("front-end embeddings -> HIP sample vs oracle, closed loop: %d of %d tokens differ" % (n_diff, otoks.numel()))
    if rep["near_tie"] == 0:
        assert n_diff == 0

    # --- decode through the checkpoint-loaded VQGAN ---
    img = vqgan.decode_indices(toks % vcfg["codebook_size"])
    assert img.shape == (2, 3, 128, 128) and torch.isfinite(img).all()
    with torch.no_grad():
        iref = O.vq_decode_indices(vsd, vcfg, toks.cpu() % vcfg["codebook_size"])
    np.testing.assert_allclose(img.cpu().numpy(), iref.numpy(), atol=1e-4)

    # --- the counter-based production path on the same inputs ---
    out = paella_amd.sample(model, cond, (2, 16, 16), unconditional_inputs=uncond, steps=3, renoise_steps=2, device=DEV, noise="philox", seed=11)
    assert int(out.min()) >= 0 and int(out.max()) < cfg["num_labels"]
