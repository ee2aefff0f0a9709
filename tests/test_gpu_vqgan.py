"""GPU: VQGAN encode / decode through the C ABI vs the reference's golden outputs and, at full size, the CPU oracle."""
import numpy as np
import pytest
import torch

import paella_amd
from oracle import golden_configs as G
from oracle import paella_oracle as O
from paella_amd import synth
from tests.helpers import weights_for

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("name,vc", [("vq_tiny_f4", G.VQ_TINY_F4), ("vq_tiny_f8", G.VQ_TINY_F8)])
def test_tiny_vs_reference(golden, built_lib, name, vc):
    g = golden(name)
    v = paella_amd.VQModel(**vc)
    sd = weights_for(v, vc["bottleneck_blocks"], g)
    v = v.to(DEV)
    img = torch.from_numpy(g["img"]).to(DEV)
    qe, lat, idx, loss = v.encode(img)
    np.testing.assert_allclose(lat.cpu().numpy(), g["lat"], atol=2e-5)
    mism = idx.cpu().numpy() != g["idx"]
    # integer output: exact, except where the reference-side latent is equidistant (within eps) from its two nearest codes --
    # the only place the unpinned third-party quantiser's tie-break / distance formulation can matter.  Counted and printed.
    rows = (torch.from_numpy(g["lat"]) * vc["scale_factor"]).permute(0, 2, 3, 1).reshape(-1, vc["c_latent"]).double()
    d = torch.cdist(rows, sd["vquantizer.codebook.weight"].double()).pow(2)
    top = d.topk(2, dim=1, largest=False).values
    near = ((top[:, 1] - top[:, 0]) < 1e-5).numpy().reshape(mism.shape)
    print(name, "token mismatches: %d of %d, all at nearest-code near-ties: %s (%d near-ties present)"
          % (int(mism.sum()), mism.size, not (mism & ~near).any(), int(near.sum())))
    assert not (mism & ~near).any(), "%d token(s) differ where the nearest code was unambiguous" % int((mism & ~near).sum())
    if not mism.any():
        np.testing.assert_allclose(qe.cpu().numpy(), g["qe"], atol=1e-6)
        np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-4)
    dec_i = v.decode_indices(torch.from_numpy(g["idx"]).to(DEV))
    np.testing.assert_allclose(dec_i.cpu().numpy(), g["dec_idx"], atol=5e-5)
    dec = v.decode(torch.from_numpy(g["qe"]).to(DEV))
    np.testing.assert_allclose(dec.cpu().numpy(), g["dec"], atol=5e-5)
    # round-trip property: decode(encode(x)[0]) == decode_indices(encode(x)[2])
    np.testing.assert_allclose(v.decode(qe).cpu().numpy(), v.decode_indices(idx).cpu().numpy(), atol=1e-5)
    # vquantizer call-site contract (src_distributed/train.py:155-156)
    rows = lat.permute(0, 2, 3, 1) * vc["scale_factor"]
    assert torch.equal(v.vquantizer.forward(rows, dim=-1)[-1], idx)
    assert v.vquantizer.idx2vq(idx, dim=1).shape == qe.shape


def test_full_size_f8_decode_vs_oracle(built_lib):
    """BASELINE decode: VQModel(levels=3) on a 32x32 token grid -> 256x256 px (38.8 GFLOP)."""
    vc = G.VQ_F8
    v = paella_amd.VQModel(**vc)
    sd = weights_for(v, vc["bottleneck_blocks"])
    v = v.to(DEV)
    g = torch.Generator().manual_seed(1)
    idx = torch.randint(0, vc["codebook_size"], (1, 32, 32), generator=g)
    with torch.no_grad():
        ref = O.vq_decode_indices(sd, vc, idx)
    got = v.decode_indices(idx.to(DEV)).cpu()
    assert got.shape == (1, 3, 256, 256)
    diff = (got - ref).abs().max().item()
    print("f8 decode: output std %.3f max|diff| %.3e" % (ref.std().item(), diff))
    assert diff <= 2e-4 * max(1.0, ref.abs().max().item())


def test_full_size_encode_vs_oracle(built_lib):
    """VQModel.encode at the released size (f8: levels=3, c_hidden=384) on a 256x256 image against the CPU oracle
    (src/vqgan.py:91-95): pre-quantisation latents to 5e-5, token indices exact except at nearest-code near-ties (counted)."""
    vc = G.VQ_F8
    v = paella_amd.VQModel(**vc)
    sd = weights_for(v, vc["bottleneck_blocks"])
    v = v.to(DEV)
    g = torch.Generator().manual_seed(17)
    img = torch.rand(1, 3, 256, 256, generator=g)
    with torch.no_grad():
        oq, olat, oidx, oloss = O.vq_encode(sd, vc, img)
    qe, lat, idx, loss = v.encode(img.to(DEV))
    np.testing.assert_allclose(lat.cpu().numpy(), olat.numpy(), atol=5e-5)
    mism = idx.cpu() != oidx
    rows = (olat * vc["scale_factor"]).permute(0, 2, 3, 1).reshape(-1, vc["c_latent"]).double()
    d = torch.cdist(rows, sd["vquantizer.codebook.weight"].double()).pow(2)
    top = d.topk(2, dim=1, largest=False).values
    near = ((top[:, 1] - top[:, 0]) < 1e-5).view(mism.shape)
    print("f8 encode 256 px: %d / %d tokens differ, all at nearest-code near-ties: %s" % (int(mism.sum()), mism.numel(), not bool((mism & ~near).any())))
    assert not (mism & ~near).any()
    if not mism.any():
        np.testing.assert_allclose(qe.cpu().numpy(), oq.numpy(), atol=1e-6)
        np.testing.assert_allclose(float(loss), float(oloss), rtol=1e-3)
    # VectorQuantize surface: losses and idx2vq run in the HIP library too
    zq, (vq_loss, commit), ids = v.vquantizer(lat * vc["scale_factor"], dim=1)
    assert torch.equal(ids, idx) and abs(float(vq_loss) * 1.25 - float(loss)) <= 1e-4 * max(1.0, float(loss))
    np.testing.assert_allclose(v.vquantizer.idx2vq(idx, dim=1).cpu().numpy(), zq.cpu().numpy(), atol=0)


def test_inpaint_composition_vs_oracle(built_lib):
    """paella_amd.inpaint against the oracle's composition of the same reference pieces: vq_encode -> add_noise with
    the user mask -> sample(init_x, t_start < 1) with explicit noise -> vq_decode_indices, on the tiny configs."""
    cfg = G.UNET_TINY
    vc = dict(G.VQ_TINY_F8, codebook_size=cfg["num_labels"])
    m = paella_amd.Paella(**cfg)
    sd = weights_for(m, sum(cfg["blocks"]))
    m = m.to(DEV)
    vq = paella_amd.VQModel(**vc)
    vsd = weights_for(vq, vc["bottleneck_blocks"])
    vq = vq.to(DEV)
    from tests.helpers import cond_for, to_dev
    g = torch.Generator().manual_seed(4)
    B, steps, t_start = 2, 4, 0.6
    img = torch.rand(B, 3, 128, 128, generator=g)  # f8 -> 16x16 tokens
    c, u = cond_for(cfg, B, 3, 0, 1), cond_for(cfg, B, 3, 0, 2)
    mask = torch.zeros(B, 16, 16, dtype=torch.int64)
    mask[:, 4:12, 4:12] = 1
    random_x = torch.randint(0, cfg["num_labels"], (B, 16, 16), generator=g)
    noise = O.replay_torch_noise(5, (B, 16, 16), cfg["num_labels"], steps, steps - 1)
    # oracle composition
    with torch.no_grad():
        _, _, otok, _ = O.vq_encode(vsd, vc, img)
        noised, _ = O.add_noise(otok, torch.full((B,), t_start), cfg["num_labels"], mask=mask, random_x=random_x)
        t_list = [float(v) for v in torch.linspace(t_start, 0.0, steps + 1)]
        temps = [float(v) for v in torch.linspace(0.7, 0.3, steps)]
        sched = torch.linspace(8.0, 8.0, steps)
        cfgs = [(float(sched[i]), float(1 - sched[i])) for i in range(steps)]
        fwd = lambda tk, rr, **inp: O.unet_forward(sd, cfg, tk, rr, **inp)
        osamp, _ = O.sample(fwd, cfg["num_labels"], c, u, (B, 16, 16), init_x=noised, steps=steps, renoise_steps=steps - 1, temperatures=temps,
                            cfgs=cfgs, t_list=t_list, noise=noise)
        oimg = O.vq_decode_indices(vsd, vc, osamp)
    toks, out = paella_amd.inpaint(m, vq, img.to(DEV), mask, to_dev(c, DEV), to_dev(u, DEV), steps=steps, t_start=t_start, keep_known=False,
                                   random_x=random_x.to(DEV), noise=noise)
    enc_same = torch.equal(vq.encode(img.to(DEV))[2].cpu(), otok)
    same = int((toks.cpu() == osamp).sum())
    print("inpaint vs oracle composition: encode tokens identical %s, %d / %d sampled tokens identical" % (enc_same, same, osamp.numel()))
    assert enc_same
    assert same == osamp.numel()
    np.testing.assert_allclose(out.cpu().numpy(), oimg.numpy(), atol=1e-4)


def test_codebook_search_lds_kernel_is_bit_identical_to_the_row_kernel(built_lib):
    """VectorQuantize.forward (src/vqgan.py:94; stand-in semantics: argmin of |e|^2 + |x|^2 - 2 x.e, first minimum) on >= 4096 rows takes the kernel that keeps
    the 8192 x 4 codebook resident in LDS; below it the wave-per-row kernel.  Same fp32 operation sequence per (row, code) pair and same tie-break, so a large call
    must equal the concatenation of small calls bit for bit, and both must match a brute-force fp64 search except at near-ties (counted)."""
    v = paella_amd.VQModel(**G.VQ_F8)
    synth.randomize_(v, seed=0)
    v = v.to(DEV)
    g = torch.Generator().manual_seed(9)
    rows = 4096 * 3 + 37                                   # ragged tail: the last 64-row chunk is partial
    x = (torch.randn(rows, 4, generator=g) * 1.2).to(DEV)
    x[5] = v.vquantizer.codebook.weight[123].detach()      # an exact hit
    zq, _, idx = v.vquantizer.forward(x, get_losses=False)
    small = torch.cat([v.vquantizer.forward(x[i:i + 1000], get_losses=False)[2] for i in range(0, rows, 1000)])
    assert torch.equal(idx, small), "%d indices differ between the two kernels" % int((idx != small).sum())
    assert int(idx[5]) == 123
    cb = v.vquantizer.codebook.weight.detach().double().cpu()
    d = (cb * cb).sum(1)[None, :] + (x.double().cpu() ** 2).sum(1)[:, None] - 2.0 * x.double().cpu() @ cb.t()
    ref = d.argmin(1)
    mism = ref != idx.cpu()
    top = d.topk(2, dim=1, largest=False).values
    near = (top[:, 1] - top[:, 0]) < 1e-5
    print("LDS codebook search vs fp64 brute force: %d / %d indices differ, all at near-ties: %s" % (int(mism.sum()), rows, not bool((mism & ~near).any())))
    assert not (mism & ~near).any()
    assert torch.equal(zq, v.vquantizer.codebook.weight.detach()[idx])
