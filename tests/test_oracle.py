"""CPU: the oracle restatement vs the golden outputs of the reference itself (tests/golden, made by oracle/make_golden.py)."""
import numpy as np
import torch

from oracle import golden_configs as G
from oracle import paella_oracle as O
from paella_amd import synth
from tests.helpers import cond_for


def _sd_from_keys(npz, n_blocks):
    shapes = {}
    for item in npz["keys"].tolist():
        k, s = item.split(":")
        shapes[k] = torch.empty([int(d) for d in s.split(",")] if s else [])
    sd = synth.synth_state_dict(shapes, seed=G.WEIGHT_SEED, n_blocks=n_blocks)
    np.testing.assert_allclose(np.array(synth.checksum(sd)), npz["checksum"], rtol=1e-9)
    return sd


def test_unet_tiny_forward_and_embeddings(golden):
    g = golden("unet_tiny_forward")
    cfg = G.UNET_TINY
    sd = _sd_from_keys(g, sum(cfg["blocks"]))
    x, r = torch.from_numpy(g["x"]), torch.from_numpy(g["r"])
    c = cond_for(cfg, 2, 5, 1, G.COND_SEED)
    taps = {}
    with torch.no_grad():
        out = O.unet_forward(sd, cfg, x, r, **c, taps=taps)
    np.testing.assert_allclose(out.numpy(), g["logits"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(taps["r_embed"].numpy(), g["r_embed"], atol=1e-6)
    np.testing.assert_allclose(taps["c_embed"].numpy(), g["c_embed"], atol=1e-5)
    ref = torch.from_numpy(g["logits"])
    top = ref.topk(2, dim=1).values
    mism = out.argmax(1) != ref.argmax(1)
    assert not (mism & ((top[:, 0] - top[:, 1]) >= 1e-4)).any(), "oracle argmax differs from the reference's away from near-ties"
    # conditioning variants: text-only and CLIP-only (byt5 of length 0)
    c2 = cond_for(cfg, 2, 3, 0, G.COND_SEED + 1)
    with torch.no_grad():
        np.testing.assert_allclose(O.unet_forward(sd, cfg, x, r, **c2).numpy(), golden("unet_tiny_forward_textonly")["logits"], atol=2e-5, rtol=1e-5)
        c3 = dict(c2, byt5=c2["byt5"][:, :0])
        np.testing.assert_allclose(O.unet_forward(sd, cfg, x, r, **c3).numpy(), golden("unet_tiny_forward_cliponly")["logits"], atol=2e-5, rtol=1e-5)


def test_unet_attn_weights_and_image_list(golden):
    g = golden("unet_tiny_attnw")
    cfg = G.UNET_TINY
    sd = _sd_from_keys(golden("unet_tiny_forward"), sum(cfg["blocks"]))
    x, r = torch.from_numpy(g["x"]), torch.from_numpy(g["r"])
    c = cond_for(cfg, 2, 5, 2, G.COND_SEED)
    with torch.no_grad():
        a = O.unet_forward(sd, cfg, x, r, **c, attn_weights=torch.from_numpy(g["attn_weights"]))
        b = O.unet_forward(sd, cfg, x, r, **c)
    np.testing.assert_allclose(a.numpy(), g["logits"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(b.numpy(), g["logits_noaw"], atol=2e-5, rtol=1e-5)
    assert np.abs(g["logits"] - g["logits_noaw"]).max() > 1e-3  # the weights do something


def test_unet_mid_and_variant(golden):
    g = golden("unet_mid_forward")
    cfg = G.UNET_MID
    sd = _sd_from_keys(g, sum(cfg["blocks"]))
    c = cond_for(cfg, 1, 0, 0, G.COND_SEED)
    with torch.no_grad():
        out = O.unet_forward(sd, cfg, torch.from_numpy(g["x"]), torch.from_numpy(g["r"]), **c)
    np.testing.assert_allclose(out[:, :, ::2, ::2].numpy(), g["logits_sub"], atol=5e-5, rtol=1e-5)
    mism = out.argmax(1).numpy() != g["argmax"]
    assert not (mism & (g["top2_margin"] > 1e-4)).any()
    gv = golden("unet_variant_forward")
    cfgv = G.UNET_VARIANT
    sdv = _sd_from_keys(gv, sum(cfgv["blocks"]))
    cv = cond_for(cfgv, 2, 3, 1, G.COND_SEED)
    with torch.no_grad():
        outv = O.unet_forward(sdv, cfgv, torch.from_numpy(gv["x"]), torch.from_numpy(gv["r"]), **cv)
    np.testing.assert_allclose(outv.numpy(), gv["logits"], atol=2e-5, rtol=1e-5)


def test_add_noise_bit_exact(golden):
    g = golden("add_noise")
    x, t = torch.from_numpy(g["x"]), torch.from_numpy(g["t"])
    torch.manual_seed(11)
    xn, mask = O.add_noise(x, t, G.UNET_TINY["num_labels"])
    assert np.array_equal(xn.numpy(), g["x_noised"]) and np.array_equal(mask.numpy(), g["mask"])
    xn2, m2 = O.add_noise(x, t, 64, mask=torch.from_numpy(g["user_mask"]), random_x=torch.from_numpy(g["random_x"]))
    assert np.array_equal(xn2.numpy(), g["x_noised_user"]) and np.array_equal(m2.numpy(), g["mask_user"])


def test_sample_loop_bit_exact(golden):
    """The restated loop + replayed torch noise reproduces the reference's sample() token-for-token (BASELINE config 1)."""
    g = golden("sample_tiny")
    cfg = G.UNET_TINY
    sd = _sd_from_keys(golden("unet_tiny_forward"), sum(cfg["blocks"]))
    cs, us = cond_for(cfg, 1, 4, 0, G.COND_SEED), cond_for(cfg, 1, 4, 0, G.COND_SEED + 5)
    noise = O.replay_torch_noise(G.SAMPLER_SEED, (1, 32, 32), cfg["num_labels"], 8, 7)
    t_list = [float(v) for v in torch.linspace(1.0, 0.0, 9)]
    temps = [float(v) for v in torch.linspace(1.0, 0.2, 8)]
    cf = (float(torch.tensor(8.0)), float(torch.tensor(1.0 - 8.0)))
    fwd = lambda tk, rr, **inp: O.unet_forward(sd, cfg, tk, rr, **inp)
    with torch.no_grad():
        toks, traj = O.sample(fwd, cfg["num_labels"], cs, us, (1, 32, 32), steps=8, renoise_steps=7, temperatures=temps,
                              cfgs=[cf] * 8, t_list=t_list, noise=noise)
    assert np.array_equal(toks.numpy(), g["tokens"])
    assert np.array_equal(torch.stack(traj).numpy(), g["traj"])


def test_vqgan(golden):
    for name, vc in (("vq_tiny_f4", G.VQ_TINY_F4), ("vq_tiny_f8", G.VQ_TINY_F8)):
        g = golden(name)
        sd = _sd_from_keys(g, vc["bottleneck_blocks"])
        img = torch.from_numpy(g["img"])
        with torch.no_grad():
            qe, lat, idx, loss = O.vq_encode(sd, vc, img)
            np.testing.assert_allclose(lat.numpy(), g["lat"], atol=1e-5)
            assert np.array_equal(idx.numpy(), g["idx"])
            np.testing.assert_allclose(qe.numpy(), g["qe"], atol=1e-6)
            np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-5)
            np.testing.assert_allclose(O.vq_decode_indices(sd, vc, idx).numpy(), g["dec_idx"], atol=2e-5)
            np.testing.assert_allclose(O.vq_decode(sd, vc, qe).numpy(), g["dec"], atol=2e-5)
            # decode(encode(x)[0]) == decode_indices(encode(x)[2])
            np.testing.assert_allclose(g["dec"], g["dec_idx"], atol=1e-5)
