"""GPU: token grids with H != W, end to end.

The reference is fully convolutional (src/modules.py:130-134,153-156,172-183; src/vqgan.py:54-89): any grid divisible by the
down-sampling factor runs.  Everything between the token gather and the logits indexes positions -- space-to-depth / depth-to-space
/ pixel-shuffle stores, the k2s2 convolutions, the 4-phase k4s2p1 transposed convolution, depthwise 3x3 borders -- so every one
of those maps is checked here against fixtures produced by the REFERENCE on non-square grids (tests/golden/*_nonsquare.npz,
oracle/make_golden.py::make_nonsquare_goldens) and, at the 570M size, against the CPU oracle."""
import numpy as np
import pytest
import torch

import paella_amd
from oracle import golden_configs as G
from oracle import paella_oracle as O
from tests.helpers import argmax_report, assert_token_parity, cond_for, stepwise_token_parity, to_dev, weights_for

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def tiny_sd(golden, built_lib):
    m = paella_amd.Paella(**G.UNET_TINY)
    sd = weights_for(m, sum(G.UNET_TINY["blocks"]), golden("unet_tiny_forward_nonsquare"))
    return m.to(DEV), sd


def _oracle_fwd(sd, cfg):
    def fwd(tk, rr, **inp):
        with torch.no_grad():
            return O.unet_forward(sd, cfg, tk, rr, **inp)
    return fwd


@pytest.mark.parametrize("tag,shape,seed", [("wide", (2, 16, 32), 17), ("tall", (1, 24, 8), 18)])
def test_tiny_forward_vs_reference(golden, tiny_sd, tag, shape, seed):
    m, _ = tiny_sd
    g = golden("unet_tiny_forward_nonsquare")
    x, r = torch.from_numpy(g[tag + "_x"]).to(DEV), torch.from_numpy(g[tag + "_r"]).to(DEV)
    assert tuple(x.shape) == shape
    c = to_dev(cond_for(G.UNET_TINY, shape[0], 5, 1, G.COND_SEED + seed), DEV)
    out = m(x, r, **c).float().cpu()
    ref = torch.from_numpy(g[tag + "_logits"])
    assert out.shape == ref.shape == (shape[0], 64, shape[1], shape[2])
    diff = (out - ref).abs().max().item()
    clear, near, n_near = argmax_report(ref, out)
    print("tiny forward %s %s vs reference: max|diff| %.3e, argmax mismatches clear=%d near-tie=%d (of %d near-tie positions)" % (tag, shape, diff, clear, near, n_near))
    assert diff <= 2e-4 and clear == 0


def test_sample_reproduces_reference_tokens_16x32(golden, tiny_sd):
    """The reference's own sample() on a 16x32 grid (src/utils.py:35-55), torch noise replayed: teacher-forced per-step parity, then the closed loop."""
    m, sd = tiny_sd
    g = golden("sample_tiny_nonsquare")
    cfg = G.UNET_TINY
    shape = (1, 16, 32)
    c, u = cond_for(cfg, 1, 4, 0, G.COND_SEED), cond_for(cfg, 1, 4, 0, G.COND_SEED + 5)
    cs, us = to_dev(c, DEV), to_dev(u, DEV)
    noise = O.replay_torch_noise(G.SAMPLER_SEED + 7, shape, cfg["num_labels"], 8, 7)
    rep = stepwise_token_parity(m, _oracle_fwd(sd, cfg), cfg["num_labels"], c, u, cs, us, noise, 8, 7, (1.0, 0.2), 8.0)
    assert_token_parity(rep, "tiny categorical, 8 steps, 16x32 grid")
    toks = paella_amd.sample(m, cs, shape, unconditional_inputs=us, steps=8, renoise_steps=7, temperature=(1.0, 0.2), cfg=8.0, device=DEV, noise=noise)
    same = int((toks.cpu().numpy() == g["tokens"]).sum())
    print("16x32 closed loop: %d / %d tokens identical to the reference" % (same, g["tokens"].size))
    if rep["near_tie"] == 0:
        assert same == g["tokens"].size


@pytest.mark.parametrize("shape", [(2, 16, 32), (3, 24, 8)])
def test_fused_tail_graph_and_shard_on_nonsquare_grids(tiny_sd, shape):
    """Counter-based noise on H != W: fused head + tail == the two-kernel path bit for bit; a captured graph == eager; a batch shard == those rows of
    the unsharded call (global-row keyed Philox: the row offset is lo * H * W)."""
    m, _ = tiny_sd
    cfg = G.UNET_TINY
    B, H, W = shape
    cs, us = to_dev(cond_for(cfg, B, 3, 0, 1), DEV), to_dev(cond_for(cfg, B, 3, 0, 2), DEV)
    kw = dict(unconditional_inputs=us, steps=4, renoise_steps=3, device=DEV, noise="philox", seed=11)
    fused = paella_amd.sample(m, cs, shape, **kw)
    unfused = paella_amd.sample(m, cs, shape, fused_tail=False, **kw)
    assert torch.equal(fused, unfused)
    assert int(fused.min()) >= 0 and int(fused.max()) < cfg["num_labels"]
    gs = paella_amd.GraphSampler(m, cs, us, shape, steps=4, renoise_steps=3, device=DEV)
    assert torch.equal(gs(cs, us, seed=11).clone(), fused)
    # rows [1, B) as a shard of the global batch
    sl = lambda d: {k: (v[1:] if v is not None else None) for k, v in d.items()}
    part = paella_amd.sample(m, sl(cs), (B - 1, H, W), unconditional_inputs=sl(us), steps=4, renoise_steps=3, device=DEV, noise="philox", seed=11, shard=(1, B))
    assert torch.equal(part, fused[1:])
    gs2 = paella_amd.GraphSampler(m, sl(cs), sl(us), (B - 1, H, W), steps=4, renoise_steps=3, device=DEV)
    assert torch.equal(gs2(sl(cs), sl(us), seed=11, shard=(1, B)), fused[1:])


@pytest.mark.parametrize("name,vc,px", [("vq_tiny_f4_nonsquare", G.VQ_TINY_F4, (64, 128)), ("vq_tiny_f8_nonsquare", G.VQ_TINY_F8, (128, 256))])
def test_vqgan_vs_reference(golden, built_lib, name, vc, px):
    """VQModel(levels=2 / 3) encode / decode / decode_indices on a 1:2 image against the reference's outputs (src/vqgan.py:91-107)."""
    g = golden(name)
    v = paella_amd.VQModel(**vc)
    sd = weights_for(v, vc["bottleneck_blocks"], g)
    v = v.to(DEV)
    img = torch.rand(1, 3, px[0], px[1], generator=torch.Generator().manual_seed(6))
    np.testing.assert_allclose(float(img.double().sum()), float(g["img_sum"]), rtol=1e-12, err_msg="the seeded test image differs from the fixture's")
    qe, lat, idx, loss = v.encode(img.to(DEV))
    assert lat.shape == g["lat"].shape and idx.shape == g["idx"].shape
    np.testing.assert_allclose(lat.cpu().numpy(), g["lat"], atol=2e-5)
    mism = idx.cpu().numpy() != g["idx"]
    rows = (torch.from_numpy(g["lat"]) * vc["scale_factor"]).permute(0, 2, 3, 1).reshape(-1, vc["c_latent"]).double()
    d = torch.cdist(rows, sd["vquantizer.codebook.weight"].double()).pow(2)
    top = d.topk(2, dim=1, largest=False).values
    near = ((top[:, 1] - top[:, 0]) < 1e-5).numpy().reshape(mism.shape)
    print(name, "token mismatches: %d of %d, all at nearest-code near-ties: %s (%d near-ties present)" % (int(mism.sum()), mism.size, not (mism & ~near).any(), int(near.sum())))
    assert not (mism & ~near).any()
    if not mism.any():
        np.testing.assert_allclose(qe.cpu().numpy(), g["qe"], atol=1e-6)
        np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-4)
    np.testing.assert_allclose(v.decode_indices(torch.from_numpy(g["idx"]).to(DEV)).cpu().numpy(), g["dec_idx"], atol=5e-5)
    np.testing.assert_allclose(v.decode(torch.from_numpy(g["qe"]).to(DEV)).cpu().numpy(), g["dec"], atol=5e-5)


def test_inpaint_on_a_nonsquare_image_vs_oracle(built_lib):
    """paella_amd.inpaint on a 64x128 px image (f8 -> 8x16 tokens) against the oracle's composition of the same reference pieces."""
    cfg = G.UNET_TINY
    vc = dict(G.VQ_TINY_F8, codebook_size=cfg["num_labels"])
    m = paella_amd.Paella(**cfg)
    sd = weights_for(m, sum(cfg["blocks"]))
    m = m.to(DEV)
    vq = paella_amd.VQModel(**vc)
    vsd = weights_for(vq, vc["bottleneck_blocks"])
    vq = vq.to(DEV)
    g = torch.Generator().manual_seed(4)
    B, steps, t_start, H, W = 2, 4, 0.6, 8, 16
    img = torch.rand(B, 3, H * 8, W * 8, generator=g)
    c, u = cond_for(cfg, B, 3, 0, 1), cond_for(cfg, B, 3, 0, 2)
    mask = torch.zeros(B, H, W, dtype=torch.int64)
    mask[:, 2:6, 3:13] = 1
    random_x = torch.randint(0, cfg["num_labels"], (B, H, W), generator=g)
    noise = O.replay_torch_noise(5, (B, H, W), cfg["num_labels"], steps, steps - 1)
    with torch.no_grad():
        _, _, otok, _ = O.vq_encode(vsd, vc, img)
        noised, _ = O.add_noise(otok, torch.full((B,), t_start), cfg["num_labels"], mask=mask, random_x=random_x)
        t_list = [float(v) for v in torch.linspace(t_start, 0.0, steps + 1)]
        temps = [float(v) for v in torch.linspace(0.7, 0.3, steps)]
        sched = torch.linspace(8.0, 8.0, steps)
        cfgs = [(float(sched[i]), float(1 - sched[i])) for i in range(steps)]
        fwd = lambda tk, rr, **inp: O.unet_forward(sd, cfg, tk, rr, **inp)
        osamp, _ = O.sample(fwd, cfg["num_labels"], c, u, (B, H, W), init_x=noised, steps=steps, renoise_steps=steps - 1, temperatures=temps, cfgs=cfgs,
                            t_list=t_list, noise=noise)
        oimg = O.vq_decode_indices(vsd, vc, osamp)
    toks, out = paella_amd.inpaint(m, vq, img.to(DEV), mask, to_dev(c, DEV), to_dev(u, DEV), steps=steps, t_start=t_start, keep_known=False,
                                   random_x=random_x.to(DEV), noise=noise)
    assert torch.equal(vq.encode(img.to(DEV))[2].cpu(), otok)
    same = int((toks.cpu() == osamp).sum())
    print("non-square inpaint vs oracle composition: %d / %d sampled tokens identical" % (same, osamp.numel()))
    assert same == osamp.numel()
    np.testing.assert_allclose(out.cpu().numpy(), oimg.numpy(), atol=1e-4)


def test_570m_forward_32x64_vs_oracle(built_lib):
    """The 573M-class stand-in on a 32x64 token grid (256x512 px) against the CPU oracle: level-1 attention 16x32 queries, level-2 8x16."""
    cfg = G.UNET_570M
    m = paella_amd.Paella(**cfg)
    sd = weights_for(m, sum(cfg["blocks"]))
    m = m.to(DEV)
    g = torch.Generator().manual_seed(23)
    x = torch.randint(0, 8192, (1, 32, 64), generator=g)
    r = torch.tensor([0.45])
    c = cond_for(cfg, 1, 3, 0, G.COND_SEED + 23)
    with torch.no_grad():
        ref = O.unet_forward(sd, cfg, x, r, **c)
    got = m(x.to(DEV), r.to(DEV), **to_dev(c, DEV)).float().cpu()
    diff, std = (got - ref).abs().max().item(), ref.std().item()
    clear, near, n_near = argmax_report(ref, got)
    print("570M 32x64: logit std %.3f, max|diff| %.3e, argmax mismatches clear=%d near-tie=%d (of %d near-tie positions / 2048)" % (std, diff, clear, near, n_near))
    assert std > 0.05 and diff <= 1e-3 * max(1.0, std) and clear == 0
    # and the full-size VQGAN decode of a 32x64 grid
    vc = G.VQ_F8
    v = paella_amd.VQModel(**vc)
    vsd = weights_for(v, vc["bottleneck_blocks"])
    v = v.to(DEV)
    idx = torch.randint(0, vc["codebook_size"], (1, 32, 64), generator=g)
    with torch.no_grad():
        iref = O.vq_decode_indices(vsd, vc, idx)
    img = v.decode_indices(idx.to(DEV)).cpu()
    assert img.shape == (1, 3, 256, 512)
    d2 = (img - iref).abs().max().item()
    print("f8 decode 32x64 tokens: max|diff| %.3e" % d2)
    assert d2 <= 2e-4 * max(1.0, iref.abs().max().item())
